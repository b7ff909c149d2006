"""20 env steps at the bench configuration (for rocprofv3 --pmc passes over k_action / k_term_rew / k_obs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_env
dev = torch.device("cuda:0")
N = int(os.environ.get("IMX_PMC_N", "4096"))
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", N, dev, 42, 4, (10, 20))
env.reset()
act = torch.randn(N, env.plan.action_dim, device=dev).clamp_(-3, 3)
for _ in range(20):
    env.step(act)
torch.cuda.synchronize()
