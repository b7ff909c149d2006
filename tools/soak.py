"""Soak: N PPO iterations of the bench configuration; checks that memory does not grow, that nothing becomes NaN and prints timing drift."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch

from bench import build_env
from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
full = "--full-step" in sys.argv  # the env owns every producer around the physics step (bench.py --full-step)
dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20), full_step=full)
venv = RslRlVecEnvWrapper(env, clip_actions=fx["agent"].get("clip_actions"))
runner = OnPolicyRunner(venv, fx["agent"], log_dir=None, device=str(dev), use_graph=True)
runner.train_mode()
alg = runner.alg
marks = []
for it in range(n_it):
    if it in (10, n_it // 2, n_it - 1):
        torch.cuda.synchronize()
        marks.append((it, time.perf_counter(), torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev)))
    runner.collect()
    with torch.inference_mode():
        alg.compute_returns(runner.last_obs)
    alg.update()
torch.cuda.synchronize()
t_end = time.perf_counter()
s = alg.loss_dict()
print("loss stats", {k: round(v, 5) for k, v in s.items()}, "lr", alg.learning_rate)
assert all(v == v for v in s.values()) and bool(torch.isfinite(alg.bucket.flat).all())
if full:
    u = env.unwrapped if hasattr(env, "unwrapped") else env
    lv = u.terrain_importer.terrain_levels
    assert int(lv.min()) >= 0 and int(lv.max()) < u.terrain_importer.max_terrain_level and bool(torch.isfinite(u.command_term.vel_command_b).all())
    assert bool(torch.isfinite(u.actuator_net.sea_hidden_state).all()) and bool(torch.isfinite(u.contact_sensor.data.current_air_time).all())
    print("full step: mean terrain level", float(lv.float().mean()), "mean |command|", float(u.command_term.vel_command_b.abs().mean()))
(i0, t0, a0, r0), (i1, t1, a1, r1), (i2, t2, a2, r2) = marks
print(f"ms/iteration: first half {(t1 - t0) / (i1 - i0) * 1e3:.3f}, second half {(t2 - t1) / (i2 - i1) * 1e3:.3f}")
print(f"memory allocated MB: {a0 / 2**20:.1f} -> {a1 / 2**20:.1f} -> {a2 / 2**20:.1f}; reserved {r0 / 2**20:.1f} -> {r2 / 2**20:.1f}")
assert a2 <= a0 * 1.01 + 2**20, "allocated memory grows"
print("soak ok:", n_it, "iterations, update mode", "graph" if alg._update_g is not None else "eager")
