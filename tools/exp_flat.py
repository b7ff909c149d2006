import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_env
from tools.exp_obs_util import timeit
dev = torch.device("cuda:0")
for task in ("Isaac-Velocity-Flat-Anymal-C-v0", "Isaac-Cartpole-v0"):
    fx, env, _ = build_env(task, 4096, dev, 42, 4, (2, 2))
    env.reset()
    print(task, "k_obs %.1f us" % timeit(env._compute_observations))
    env.plan.enable_corruption = False
    print(task, "k_obs no-noise %.1f us" % timeit(env._compute_observations))
