import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("IMX_EXP_LIB"):
    from isaaclab_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "tools", "libimx_%s.so" % os.environ["IMX_EXP_LIB"])
import torch
from bench import build_env
from tools.exp_obs_util import timeit
dev = torch.device("cuda:0")
for N in (4096, 65536):
    fx, env, _ = build_env("Isaac-Velocity-Flat-Anymal-C-v0", N, dev, 42, 4, (10, 20))
    env.reset()
    print("flat task obs N=%d  %.2f us" % (N, timeit(lambda: env._compute_observations(frame_current=True))))
