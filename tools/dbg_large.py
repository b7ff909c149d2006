import os, sys
sys.path.insert(0, "/root/repo")
import torch
from bench import build_env, time_obs_kernel
dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
env.reset()
print("4096:", time_obs_kernel(env) * 1e6)
for snaps, seed in ((1, 7), (4, 7), (1, 42)):
    _, eb, _ = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 65536, dev, seed, snaps, (10, 20), mesh=env.terrain)
    eb.reset()
    print("65536 snaps", snaps, "seed", seed, time_obs_kernel(eb, launches=50) * 1e6)
    a = torch.zeros(65536, 12, device=dev)
    eb.step(a)
    print("   after a step:", time_obs_kernel(eb, launches=50) * 1e6)
    del eb
