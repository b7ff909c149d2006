"""Per-wave timeline of k_obs<false> (bench configuration): builds tools/libimx_trace.so (-DIMX_TRACE), runs the env kernels and
prints when waves start and end, how long they live and how many are resident.  `python tools/trace_kobs.py build` (here, hipcc)
then `python tools/trace_kobs.py` on the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libimx_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from isaaclab_amd import build as b

    cmd = ["/opt/rocm/bin/hipcc", *b.FLAGS, "-DIMX_TRACE", *[os.path.join(b.CSRC, s) for s in b.SOURCES], "-o", LIB]
    subprocess.check_call(cmd)
    print(LIB)
    sys.exit(0)

import ctypes

import numpy as np
import torch

from isaaclab_amd import _lib

_lib.LIB_PATH = LIB
from bench import build_env

dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
env.reset()
L = _lib.lib()
L.imx_debug_trace.argtypes = [ctypes.c_void_p]
buf = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device=dev)
for _ in range(5):
    env._compute_observations()
torch.cuda.synchronize()
assert L.imx_debug_trace(buf.data_ptr()) == 0
env._compute_observations()
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(4096, 4, 8)[:, :3]  # 3 waves per block
t0, t1 = t[..., 0].astype(np.float64), t[..., 1].astype(np.float64)
base = t0.min()
s, e = (t0 - base) * 0.01, (t1 - base) * 0.01  # us (100 MHz)
life = e - s
print(f"waves {s.size}; first start 0, last start {s.max():.2f} us, last end {e.max():.2f} us")
print("wave life us: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f" % (
    life.mean(), *np.percentile(life, [10, 50, 90, 99]), life.max()))
for w in range(3):
    print(f"  wave {w}: mean life {life[:, w].mean():.2f} us")
tp, tr, tb = [(t[..., k].astype(np.float64) - base) * 0.01 for k in (6, 4, 5)]
print("phases (mean us per wave): start -> scanner state broadcast %.2f | -> rays done %.2f | barrier wait %.2f | columns %.2f" % (
    (tp - s).mean(), (tr - tp).mean(), (tb - tr).mean(), (e - tb).mean()))
edges = np.arange(0, e.max() + 1.0, 1.0)
print("t(us): resident waves | blocks started in this us")
bs = s.min(axis=1)
for a in edges:
    res = int(((s <= a) & (e > a)).sum())
    started = int(((bs >= a) & (bs < a + 1.0)).sum())
    print(f"  {a:5.1f}: {res:6d} | {started:5d}")
hw = t[..., 2]
cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)
xcc = t[..., 3] & 0xF
key = xcc[:, 0] * 256 + cu[:, 0]
uniq, cnt = np.unique(key, return_counts=True)
print(f"distinct (xcc, se/sh/cu) slots used: {len(uniq)}; blocks per slot min {cnt.min()} max {cnt.max()}")
# life vs start time: do later waves run faster?
order = np.argsort(bs)
q = len(order) // 4
for k in range(4):
    idx = order[k * q:(k + 1) * q]
    print(f"  blocks by start quartile {k}: start {bs[idx].mean():6.2f} us, block life {(e[idx].max(axis=1) - bs[idx]).mean():6.2f} us")
