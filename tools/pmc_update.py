"""Workload for the PMC passes over the PPO-update kernels (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`):
imx_mlp_dw / imx_mlp_head_* on the Anymal-C rough minibatch shapes, 5 launches each + a calibration copy."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd._lib import check, lib

L = lib()
dev = torch.device("cuda:0")
M = 24576
st = torch.cuda.current_stream().cuda_stream
for N, K in ((512, 235), (256, 512), (128, 256)):
    dY, X = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    nb = int(L.imx_mlp_scratch_bytes(M, N, K))
    scr = torch.empty(nb, dtype=torch.uint8, device=dev)
    for _ in range(5):
        check(L.imx_mlp_dw(M, N, K, dY.data_ptr(), N, X.data_ptr(), K, dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nb, st))
src = torch.empty(64 * 1024 * 1024, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(5):
    dst.copy_(src)
torch.cuda.synchronize()
print("done")
