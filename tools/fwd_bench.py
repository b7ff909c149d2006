"""imx_mlp_fwd_elu (first layer + ELU, one launch) against library addmm + ELU at the update's shape (experiment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd import _lib
from isaaclab_amd.rsl_rl import gemm_tuning
gemm_tuning.enable_recorded_gemm_tuning()
L = _lib.lib()

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

SHAPES = ((24576, 1024, 235, 236), (24576, 512, 235, 236), (24576, 256, 48, 48))
for M, N, K, pitch in (SHAPES[:1] if "--one" in sys.argv else SHAPES):
    X = torch.randn(M, pitch, device="cuda")[:, :K]
    W, b = torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t_f = timeit(lambda: _lib.check(L.imx_mlp_fwd_elu(M, N, K, X.data_ptr(), X.stride(0), W.data_ptr(), b.data_ptr(), 1.0, 1, Y.data_ptr(), N, st)))
    t_l = timeit(lambda: torch.nn.functional.elu(torch.addmm(b, X, W.t()), inplace=True))
    t_g = timeit(lambda: torch.addmm(b, X, W.t()))
    print(f"M={M} N={N} K={K}: fused {t_f:.1f} us ({2.0 * M * N * K / t_f / 1e6:.1f} TF) | library GEMM {t_g:.1f} + ELU = {t_l:.1f} us")
