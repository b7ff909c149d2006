"""imx_mlp_dw on the first layer (ragged K) with X rows packed (ld = K: 4-byte mover loads) against rows padded to a multiple of
four floats (ld = ceil4(K): 16-byte mover loads): same result, which is faster?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd._lib import check, current_stream, lib

dev = torch.device("cuda:0")
L = lib()
st = current_stream(dev)
M = 24576
for N, K in ((512, 235), (512, 310), (128, 48)):
    dY = torch.randn(M, N, device=dev)
    Kp = (K + 3) & ~3
    Xp = torch.randn(M, Kp, device=dev)
    Xp[:, K:] = float("nan")  # the padding must never reach a written dW column
    Xc = Xp[:, :K].contiguous()
    nb = int(L.imx_mlp_scratch_bytes(M, N, K))
    scr = torch.empty(nb, dtype=torch.uint8, device=dev)
    outs = []
    for X, ld, tag in ((Xc, K, "packed"), (Xp, Kp, "padded")):
        dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        args = (M, N, K, dY.data_ptr(), N, X.data_ptr(), ld, dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nb, st)
        for _ in range(3):
            check(L.imx_mlp_dw(*args))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            check(L.imx_mlp_dw(*args))
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) * 1e3 / 30
        outs.append(dW.clone())
        print(f"{N}x{K} {tag:7s} ld={ld}: {us:6.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s")
    ref = dY.double().t() @ Xc.double()
    e0, e1 = float((outs[0].double() - ref).abs().max()), float((outs[1].double() - ref).abs().max())
    print(f"   max abs err vs fp64: packed {e0:.3e} padded {e1:.3e}; equal={bool(torch.equal(outs[0], outs[1]))} finite={bool(torch.isfinite(outs[1]).all())}")
