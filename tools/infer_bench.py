"""k_mlp_infer alone (actor + critic stacks of the rough-terrain task, 4096 samples): timing, and a target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
from isaaclab_amd.rsl_rl.ppo import FusedInference, _mlp_layers, FlatParams

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
pol = ActorCritic(235, 235, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu").cuda()
FlatParams(pol)
inf = FusedInference(_mlp_layers(pol.actor), _mlp_layers(pol.critic))
x = torch.randn(M, 235, device="cuda")
mu, v = torch.empty(M, 12, device="cuda"), torch.empty(M, 1, device="cuda")
junk = torch.empty(64 << 20, device="cuda")  # 256 MB written between launches: what the env kernels do to the caches
for cold in (False, True):
    for _ in range(5):
        inf(x, mu, v)
    ts = []
    for _ in range(30):
        if cold:
            junk.fill_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); inf(x, mu, v); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    print(f"M={M} {'cold' if cold else 'warm'} caches: median {ts[len(ts)//2]:.1f} us, min {ts[0]:.1f} us")
