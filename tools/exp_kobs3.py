"""k_obs variants (HIP events, back-to-back launches): where the 30 us go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_env
from tools.exp_obs_util import timeit

dev = torch.device("cuda:0")
NENV = int(os.environ.get("IMX_EXP_N", "4096"))
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", NENV, dev, 42, 4, (10, 20))
env.reset()
def f():
    env._compute_observations(frame_current=True)


def moved():  # the frame table (positions) is refreshed by k_frame only without frame_current
    env._compute_observations()

print("mesh", env.terrain.nx, env.terrain.ny, "lattice cells", env.terrain.num_lattice_cells, "general cells", env.terrain.num_general_cells, "of which flat", env.terrain.num_flat_cells)
print("normal                      %.1f us" % timeit(f))
env.plan.enable_corruption = False
print("no noise                    %.1f us" % timeit(f))
env.plan.enable_corruption = True
pos = env.feed._stack["root_pos_w"]
saved = pos.clone()
pos[:, :, 0] = 3.137; pos[:, :, 1] = -2.211
moved()
print("same position (cache-hot)   %.1f us" % timeit(f))
pos.copy_(saved)
moved()
# envs only over height-field tiles (lattice cells) / only over box tiles: tile kinds cycle with period 10 along the column index
import numpy as np
for name, kinds in (("lattice-only envs", (3, 7, 8, 9)), ("general-only envs", (0, 1, 2, 4, 5, 6))):
    p = saved.clone()
    N = p.shape[1]
    g = torch.Generator().manual_seed(1)
    r = torch.randint(0, 10, (N,), generator=g)
    c = torch.tensor(kinds)[torch.randint(0, len(kinds), (N,), generator=g)] + 10 * torch.randint(0, 2, (N,), generator=g)
    # tile (r, c): index r*20 + c, kind = index % 10 = c % 10 -> centre of the tile
    x = (-40.0 + (r.float() + 0.5) * 8.0 + (torch.rand(N, generator=g) - 0.5) * 5.0).to(dev)
    y = (-80.0 + (c.float() + 0.5) * 8.0 + (torch.rand(N, generator=g) - 0.5) * 5.0).to(dev)
    p[:, :, 0] = x; p[:, :, 1] = y
    pos.copy_(p)
    moved()
    print("%-27s %.1f us" % (name, timeit(f)))
pos.copy_(saved)
fx2, env2, _ = build_env("Isaac-Velocity-Flat-Anymal-C-v0", NENV, dev, 42, 4, (10, 20))
env2.reset()
print("flat task (D=48, no rays)   %.1f us" % timeit(lambda: env2._compute_observations(frame_current=True)))
