"""Experiment: where does k_obs spend its time?  (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_env

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
env.reset()
print("mesh", env.terrain.nx, env.terrain.ny, "general refs", env.terrain.num_refs, "max", env.terrain.max_refs, "lattice cells", env.terrain.num_lattice_cells, "general cells", env.terrain.num_general_cells)
print("k_obs normal            %.1f us" % timeit(env._compute_observations))
env.plan.enable_corruption = False
print("k_obs no noise          %.1f us" % timeit(env._compute_observations))
# all robots on the same spot -> perfect locality
pos = env.feed._stack["root_pos_w"]
saved = pos.clone()
pos[:, :, 0] = 3.137; pos[:, :, 1] = -2.211
print("k_obs same position     %.1f us" % timeit(env._compute_observations))
pos.copy_(saved)
# robots sorted along x-major cell order -> neighbouring waves touch neighbouring memory
# standalone ray-cast of the same number of rays, thread per ray
N, R = 4096, 187
starts = torch.empty(N, R, 3, device=dev)
starts[..., 0] = (torch.rand(N, 1, device=dev) * 70 - 35) + torch.linspace(-0.8, 0.8, R, device=dev)
starts[..., 1] = (torch.rand(N, 1, device=dev) * 150 - 75) + torch.rand(N, R, device=dev)
starts[..., 2] = 20.6
dirs = torch.zeros(N, R, 3, device=dev); dirs[..., 2] = -1
print("k_raycast 766k rays     %.1f us" % timeit(lambda: env.terrain.raycast(starts, dirs)))
fx2, env2, _ = build_env("Isaac-Velocity-Flat-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
env2.reset()
print("k_obs flat (D=48)       %.1f us" % timeit(env2._compute_observations))
a = torch.zeros(4096, 12, device=dev)
env.step(a)
print("env.step rough          %.1f us (host+3 kernels)" % timeit(lambda: env.step(a)))
print("k_term_rew rough        %.1f us" % timeit(lambda: env._lib.imx_terminations_rewards(env._plan_h, 4096, __import__('ctypes').byref(env._state()), __import__('ctypes').byref(env._bufs), torch.cuda.current_stream().cuda_stream)))
t0 = time.perf_counter()
for _ in range(1000): env._state()
print("host _state() %.2f us" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
for _ in range(200): env.step(a)
print("host env.step() issue %.2f us" % ((time.perf_counter() - t0) * 1e6 / 200)); torch.cuda.synchronize()
