"""k_obs on (a) the bench terrain, (b) a pure lattice terrain (no snapped vertices, no border)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tools.exp_obs_util import timeit
from isaaclab_amd.env import ManagerBasedRLEnv, load_task_cfg
from isaaclab_amd.robots import ROBOTS
from isaaclab_amd.state_feed import StateFeed
from isaaclab_amd.terrain import height_field_to_mesh, make_rough_terrain
dev = torch.device("cuda:0")
fx = load_task_cfg("Isaac-Velocity-Rough-Anymal-C-v0")
robot = ROBOTS[fx["robot"]]
rng = np.random.default_rng(0)
hf = np.rint(rng.uniform(0, 10, size=(801, 1601)))
v, t = height_field_to_mesh(hf, 0.1, 0.005, None)
v[:, 0] -= 40; v[:, 1] -= 80
for name, terrain, ext in (("lattice-only", (v, t), (38.0, 78.0)),):
    feed = StateFeed(robot, 4096, dev, seed=42, num_snapshots=4, extent_xy=ext)
    env = ManagerBasedRLEnv(fx, state_feed=feed, terrain=terrain, terrain_cell=0.1)
    env.reset()
    tm = env.terrain
    print(name, "lattice cells", tm.num_lattice_cells, "general cells", tm.num_general_cells, "general recs", tm.num_refs)
    print(name, "k_obs %.1f us" % timeit(env._compute_observations))
    env.plan.enable_corruption = False
    print(name, "k_obs no-noise %.1f us" % timeit(env._compute_observations))
    env.materialize_ray_hits = True
    print(name, "k_obs +hits out %.1f us" % timeit(env._compute_observations))
