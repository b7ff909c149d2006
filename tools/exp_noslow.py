"""Timing-only experiment: the env step kernels with every rare ray path compiled out (tools/libimx_noslow.so, -DIMX_EXP_NOSLOW: no
neighbour-cell visits for rays within tau of a cell boundary, GENERAL cells answer "miss") -- how much of k_obs is the slow paths?
Results are WRONG by construction; only the time is of interest.  Build: see the hipcc line in DESIGN.md section 4 / tools/trace_kobs.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaaclab_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libimx_%s.so" % os.environ.get("IMX_EXP_LIB", "noslow"))
import runpy
sys.argv = ["step_bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "tools", "step_bench.py"), run_name="__main__")
