"""imx_mlp_infer vs the per-layer library path at rollout size (experiment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd.rsl_rl import gemm_tuning
gemm_tuning.enable_recorded_gemm_tuning()
from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
from isaaclab_amd.rsl_rl.ppo import FusedInference, _mlp_layers, mlp_forward

def timeit(fn, n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for M, D in ((4096, 235), (2048, 235), (1024, 235), (16384, 235)):
    pol = ActorCritic(D, D, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128]).cuda()
    x = torch.randn(M, D, device="cuda")
    la, lc = _mlp_layers(pol.actor), _mlp_layers(pol.critic)
    inf = FusedInference(la, lc)
    inf1 = FusedInference(la)
    mu, val = torch.empty(M, 12, device="cuda"), torch.empty(M, 1, device="cuda")
    with torch.inference_mode():
        t_f = timeit(lambda: inf(x, mu, val))
        t_f1 = timeit(lambda: inf1(x, mu))
        t_l = timeit(lambda: (mlp_forward(la, x), mlp_forward(lc, x)))
    fl = 2.0 * M * sum(l.in_features * l.out_features for l, _ in la + lc)
    print(f"M={M} D={D}: fused both {t_f:.1f} us ({fl / t_f / 1e6:.1f} TF), fused actor only {t_f1:.1f} us, library (serial, one stream) {t_l:.1f} us")
