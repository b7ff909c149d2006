"""Output layer of the update: imx_mlp_head_fwd_bwd (one launch) against imx_mlp_head_fwd_loss + imx_mlp_head_bwd (experiment)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd import _lib
L = _lib.lib()
M, K = 24576, 128
for A in (12, 1):
    g = torch.Generator().manual_seed(A)
    z = torch.randn(M, K, generator=g).cuda()
    W, b = (0.1 * torch.randn(A, K, generator=g)).cuda(), (0.1 * torch.randn(A, generator=g)).cuda()
    sigma = (0.5 + torch.rand(A, generator=g)).cuda()
    act = torch.randn(M, A, generator=g).cuda()
    old_logp = (-1.0 - torch.rand(M, generator=g) * A).cuda()
    adv, ret, old_v = torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    st = _lib.current_stream(z.device)
    y = torch.empty(M, A, device="cuda")
    dmu, dsg, dv = torch.zeros(M, A, device="cuda"), torch.zeros(M, A, device="cuda"), torch.zeros(M, 1, device="cuda")
    dprev, dW, db = torch.empty(M, K, device="cuda"), torch.empty(A, K, device="cuda"), torch.empty(A, device="cuda")
    nbytes = int(L.imx_mlp_scratch_bytes(M, A, K))
    scr = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    if A > 1:
        hl = _lib.ImxHeadLoss(mode=1, sigma_stride=0, use_clipped_value_loss=1, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.005, grad_scale=1.0,
                              sigma_d=sigma.data_ptr(), actions_d=act.data_ptr(), old_logp_d=old_logp.data_ptr(), advantages_d=adv.data_ptr(),
                              dmu_d=dmu.data_ptr(), dsigma_d=dsg.data_ptr())
    else:
        hl = _lib.ImxHeadLoss(mode=2, sigma_stride=0, use_clipped_value_loss=1, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.005, grad_scale=1.0,
                              returns_d=ret.data_ptr(), old_values_d=old_v.data_ptr(), dvalue_d=dv.data_ptr())
    zz = z.clone()

    def split():
        _lib.check(L.imx_mlp_head_fwd_loss(M, K, A, zz.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 1.0, ctypes.byref(hl), st))
        _lib.check(L.imx_mlp_head_bwd(M, K, A, (dmu if A > 1 else dv).data_ptr(), zz.data_ptr(), K, W.data_ptr(), 1.0, 1, dprev.data_ptr(),
                                      dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nbytes, st))

    def fused():
        _lib.check(L.imx_mlp_head_fwd_bwd(M, K, A, z.data_ptr(), K, W.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 1.0, ctypes.byref(hl),
                                          dprev.data_ptr(), dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nbytes, None, st))

    for name, fn in (("split (2 launches + reduce)", split), ("fused (1 launch + reduce)", fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        print(f"A={A} {name}: {a.elapsed_time(e) / 50 * 1e3:.1f} us")
