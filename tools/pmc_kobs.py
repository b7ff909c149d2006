"""k_obs only, 10 launches scattered + 10 launches same-position (for PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_env
dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
env.reset()
for _ in range(10):
    env._compute_observations()
torch.cuda.synchronize()
pos = env.feed._stack["root_pos_w"]
pos[:, :, 0] = 3.137; pos[:, :, 1] = -2.211
for _ in range(10):
    env._compute_observations()
torch.cuda.synchronize()
