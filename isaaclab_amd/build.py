"""Build libimx.so (hipcc, gfx950 only) in-tree.  ``python -m isaaclab_amd.build``."""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libimx.so")
SOURCES = ("core.hip", "mesh.hip", "step.hip", "rollout.hip", "producers.hip", "mlp.hip", "mlp_fwd.hip", "events.hip", "orchestrate.hip")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",  # keep the reference's fp32 association: no fused multiply-adds
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "imx.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libimx.so cannot be built (and no prebuilt library is present)")
    cmd = [hipcc, *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
