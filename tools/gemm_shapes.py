"""Per-shape timing of the PPO-update GEMMs exactly as isaaclab_amd/rsl_rl/ppo.py issues them (experiment, not product)."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd.rsl_rl import gemm_tuning

gemm_tuning.enable_recorded_gemm_tuning()
from isaaclab_amd._lib import lib
L = lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
dev = "cuda"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot = 0.0
totf = 0.0
for net, dims in (("actor", [235, 512, 256, 128, 12]), ("critic", [235, 512, 256, 128, 1])):
    for i in range(4):
        K, N = dims[i], dims[i + 1]
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        d = torch.randn(M, N, device=dev)
        gW = torch.empty_like(W)
        fl = 2.0 * M * K * N
        t_f = timeit(lambda: torch.addmm(b, x, W.t()))
        t_w = timeit(lambda: torch.mm(d.t(), x, out=gW))
        t_x = timeit(lambda: torch.mm(d, W)) if i > 0 else 0.0
        nb = int(L.imx_mlp_scratch_bytes(M, N, K))
        scr = torch.empty(nb, dtype=torch.uint8, device=dev)
        gb = torch.empty(N, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        t_imx = timeit(lambda: L.imx_mlp_dw(M, N, K, d.data_ptr(), N, x.data_ptr(), K, gW.data_ptr(), gb.data_ptr(), scr.data_ptr(), nb, st))
        extra = f" | imx_dw+db {t_imx:7.1f} us {fl / t_imx / 1e6:6.1f} TF"
        hh = torch.nn.functional.elu(torch.randn(M, N, device=dev))
        dz = torch.empty(M, N, device=dev)
        t_elu = timeit(lambda: torch.ops.aten.elu_backward(d, 1.0, 1.0, 1.0, True, hh))
        t_fused = timeit(lambda: L.imx_mlp_dw_elu(M, N, K, d.data_ptr(), N, hh.data_ptr(), N, 1.0, dz.data_ptr(), N, x.data_ptr(), K,
                                                  gW.data_ptr(), gb.data_ptr(), scr.data_ptr(), nb, st))
        extra += f" | elu_bwd {t_elu:5.1f} us, imx_dw_elu {t_fused:6.1f} us"
        if N <= 16:
            y = torch.empty(M, N, device=dev)
            dp = torch.empty(M, K, device=dev)
            t_hf = timeit(lambda: L.imx_mlp_head_fwd(M, K, N, x.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), 0, 0.0, st))
            t_hb = timeit(lambda: L.imx_mlp_head_bwd(M, K, N, d.data_ptr(), x.data_ptr(), K, W.data_ptr(), 1.0, 1, dp.data_ptr(), gW.data_ptr(),
                                                     gb.data_ptr(), scr.data_ptr(), nb, st))
            extra += f" | head fwd {t_hf:6.1f} us, head bwd (dW,db,dX,ELU') {t_hb:6.1f} us"
        print(f"{net} L{i} {K:4d}->{N:4d}: fwd {t_f:7.1f} us {fl / t_f / 1e6:6.1f} TF | dW {t_w:7.1f} us {fl / t_w / 1e6:6.1f} TF | "
              f"dX {t_x:7.1f} us {(fl / t_x / 1e6) if t_x else 0:6.1f} TF" + extra)
        tot += t_f + t_w + t_x
        totf += fl * (3 if i > 0 else 2)
print(f"sum {tot:.1f} us per minibatch (serial), {totf / tot / 1e6:.1f} TFLOP/s; x20 = {tot * 20 / 1e3:.2f} ms per iteration")
