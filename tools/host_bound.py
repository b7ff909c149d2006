"""Is PPO.update host-bound?  Host enqueue time vs device time of one update (experiment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_env, TASK
from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

dev = torch.device("cuda:0")
fx, env, ntri = build_env(TASK, 4096, dev, 42, 4, (4, 6))
venv = RslRlVecEnvWrapper(env, clip_actions=fx["agent"].get("clip_actions"))
runner = OnPolicyRunner(venv, fx["agent"], log_dir=None, device=str(dev), use_graph=True)
runner.train_mode()
for it in range(6):
    runner.collect()
    with torch.inference_mode():
        runner.alg.compute_returns(runner.last_obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.alg.update()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"iter {it}: host enqueue {1e3 * (t1 - t0):.2f} ms, until device idle {1e3 * (t2 - t0):.2f} ms")
