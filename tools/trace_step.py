"""Per-wave phase timeline of k_term_rew (bench configuration): builds tools/libimx_trace.so (-DIMX_TRACE) with 8 wall-clock stamps per
wave (start, tables staged, after barrier, items done, after barrier, rewards finished, after barrier, end).
`python tools/trace_kobs.py build` here (hipcc), then `python tools/trace_step.py` on the GPU box."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isaaclab_amd import _lib

_lib.LIB_PATH = os.path.join(ROOT, "tools", "libimx_trace.so")
from bench import build_env

N = int(os.environ.get("IMX_PMC_N", "4096"))
dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", N, dev, 42, 4, (10, 20))
env.reset()
L = _lib.lib()
L.imx_debug_trace.argtypes = [ctypes.c_void_p]
act = torch.randn(N, env.plan.action_dim, device=dev).clamp_(-3, 3)
for _ in range(5):
    env.step(act)
nb = (N + 15) // 16
buf = torch.zeros(nb * 16 * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
assert L.imx_debug_trace(buf.data_ptr()) == 0
env._process_action(act)
env.feed.advance()
_lib.check(L.imx_terminations_rewards(env._plan_h, N, ctypes.byref(env._state()), ctypes.byref(env._bufs), 1, _lib.current_stream(dev)))
torch.cuda.synchronize()
assert L.imx_debug_trace(None) == 0
t = buf.cpu().numpy().reshape(nb, 16, 8).astype(np.float64)
used = t[..., 0] > 0
base = t[..., 0][used].min()
t = (t - base) * 0.01  # us
names = ["start", "staged", "barrier0", "items done", "barrier1", "rewards done", "barrier2", "end"]
items = env.plan.termination_terms + env.plan.reward_terms
print(f"waves {int(used.sum())} in {int(used.any(axis=1).sum())} workgroups; last end {t[..., 7][used].max():.2f} us")
print("stamp            mean      p50      p99      max   (us since the first wave started)")
for k, n in enumerate(names):
    v = t[..., k][used]
    print(f"{n:14s} {v.mean():8.2f} {np.percentile(v, 50):8.2f} {np.percentile(v, 99):8.2f} {v.max():8.2f}")
print("per wave (= item): mean duration of phase 1 (barrier0 -> items done) and of phase 2 (barrier1 -> rewards done)")
for w in range(16):
    if not used[:, w].any():
        continue
    m = used[:, w]
    name = items[w].name if w < len(items) else "-"
    print(f"  wave {w:2d} {name:24s} start->staged {np.mean(t[m, w, 1] - t[m, w, 0]):6.2f}  phase1 {np.mean(t[m, w, 3] - t[m, w, 2]):6.2f}"
          f"  phase2 {np.mean(t[m, w, 5] - t[m, w, 4]):6.2f}  phase3 {np.mean(t[m, w, 7] - t[m, w, 6]):6.2f}")
