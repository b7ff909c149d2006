"""k_gather_rows timing at the bench's minibatch shape (24576 samples of a 24 x 4096 rollout, Anymal-C rough widths)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd import _lib

dev = torch.device("cuda:0")
L = _lib.lib()
T, N = 24, 4096
widths = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "235,12,1,1,1,1,12,12".split(","))]
src = [torch.randn(T * N, w, device=dev) for w in widths]
M = T * N // 4
dst = [torch.empty(M, w, device=dev) for w in widths]
idx = torch.randperm(T * N, device=dev)[:M]
n = len(widths)
sp = (ctypes.c_void_p * n)(*[s.data_ptr() for s in src])
dp = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dst])
wp = (ctypes.c_int32 * n)(*widths)
st = _lib.current_stream(dev)
for _ in range(5):
    _lib.check(L.imx_gather_rows(M, idx.data_ptr(), n, sp, dp, wp, st))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    _lib.check(L.imx_gather_rows(M, idx.data_ptr(), n, sp, dp, wp, st))
b.record()
torch.cuda.synchronize()
ok = all(torch.equal(d, s[idx]) for d, s in zip(dst, src))
byt = 2 * 4 * M * sum(widths)
us = a.elapsed_time(b) * 1e3 / 50
print(f"widths {widths}: {us:.1f} us per launch, {byt / us / 1e3:.0f} GB/s, exact={ok}")
