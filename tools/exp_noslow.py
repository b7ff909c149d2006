"""Timing-only experiment: the env step kernels with the rare ray paths compiled out -- how much of k_obs do 0.9 % of the rays cost?

    python tools/exp_noslow.py build          # here (hipcc): tools/libimx_{nogen,nonb,noslow}.so
    IMX_EXP_LIB=nogen|nonb|noslow python tools/exp_noslow.py [--num-envs ...]      # on the GPU box: tools/step_bench.py on that library

-DIMX_EXP_NOGENERAL: GENERAL cells answer "miss"; -DIMX_EXP_NONEIGHBOUR: no neighbour-cell visits for rays within tau of a cell boundary;
noslow = both.  Results are WRONG by construction; only the time is of interest."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"nogen": ["-DIMX_EXP_NOGENERAL"], "nonb": ["-DIMX_EXP_NONEIGHBOUR"], "noslow": ["-DIMX_EXP_NOGENERAL", "-DIMX_EXP_NONEIGHBOUR"]}
if sys.argv[1:2] == ["build"]:
    import isaaclab_amd.build as b
    for name, defs in VARIANTS.items():
        out = os.path.join(ROOT, "tools", f"libimx_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *b.FLAGS, *defs, *[os.path.join(b.CSRC, s) for s in b.SOURCES], "-o", out])
        print(out)
    sys.exit(0)
from isaaclab_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libimx_%s.so" % os.environ.get("IMX_EXP_LIB", "noslow"))
import runpy
sys.argv = ["step_bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "tools", "step_bench.py"), run_name="__main__")
