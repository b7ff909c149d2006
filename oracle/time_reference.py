"""TEST INFRASTRUCTURE (build container only): how far is the CPU baseline of bench.py (the restatement oracle/mdp_oracle.py, kind
"port") from the REAL reference managers on the same cores?

    python oracle/time_reference.py [--num-envs 4096] [--threads 8] [--seconds 10]

Both sides run Isaac-Velocity-Rough-Anymal-C-v0 at N envs on the same synthetic state feed (seed 42), height-scan hits supplied (no
ray-cast on either side), observation noise on (torch.rand_like), per env step:
  reference   : the real TerminationManager.compute + RewardManager.compute + ObservationManager.compute (+ ActionManager.process_action)
                imported from /root/reference through oracle/ref_import.py (simulator stubbed), driven like oracle/gen_golden.py does;
  restatement : OracleEnv.process_action + post_physics_step (what bench.py's cpu_baseline leg times on the GPU box).
Prints both rates and their ratio; BASELINE.md section 3 records the result."""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from oracle import gen_golden as gg  # installs the stubbed import of the reference
    from isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.rough_env_cfg import AnymalCRoughEnvCfg

    from isaaclab_amd.env import load_task_cfg
    from isaaclab_amd.robots import ANYMAL_C
    from isaaclab_amd.state_feed import StateFeed
    from oracle.mdp_oracle import OracleEnv

    N = args.num_envs
    g = torch.Generator().manual_seed(0)
    act = torch.randn(N, 12, generator=g).clamp_(-3, 3)
    hits = torch.rand(N, 187, 3, generator=g) * 0.2

    # ---- the real reference
    feed = StateFeed(ANYMAL_C, N, "cpu", seed=42, num_snapshots=4)
    env_cfg = AnymalCRoughEnvCfg()
    env = gg.build_ref_env(env_cfg, ANYMAL_C, feed)
    sc = env.scene.sensors["height_scanner"]
    sc.data.ray_hits_w = hits
    env.episode_length_buf[:] = torch.randint(0, env.max_episode_length, (N,), generator=g)

    def ref_step():
        env.action_manager.process_action(act)
        feed.advance()
        sc.data.pos_w = feed["root_pos_w"]
        env.episode_length_buf += 1
        reset_buf = env.termination_manager.compute()
        env.reward_manager.compute(dt=env.step_dt)
        ids = reset_buf.nonzero(as_tuple=False).squeeze(-1)
        if len(ids) > 0:
            env.observation_manager.reset(ids)
            env.action_manager.reset(ids)
            env.reward_manager.reset(ids)
            env.termination_manager.reset(ids)
            env.episode_length_buf[ids] = 0
        env.observation_manager.compute()

    # ---- the restatement (bench.py cpu_baseline, minus GAE)
    fx = load_task_cfg("Isaac-Velocity-Rough-Anymal-C-v0")
    feed2 = StateFeed(ANYMAL_C, N, "cpu", seed=42, num_snapshots=4)
    orc = OracleEnv(fx["env"], ANYMAL_C.joint_names, ANYMAL_C.body_names, N, feed2.__getitem__, feed2.gravity_dir)
    orc.ray_hits_w = hits
    orc.episode_length_buf[:] = env.episode_length_buf

    def port_step():
        orc.process_action(act)
        feed2.advance()
        orc.post_physics_step(None)

    def rate(fn):
        for _ in range(5):
            fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.seconds:
            fn()
            n += 1
        dt = time.perf_counter() - t0
        return N * n / dt, 1e3 * dt / n

    r_ref, ms_ref = rate(ref_step)
    r_port, ms_port = rate(port_step)
    print(f"threads {args.threads}, N {N}, torch {torch.__version__}")
    print(f"reference managers : {r_ref / 1e6:.3f} M env-steps/s ({ms_ref:.2f} ms per {N}-env step)")
    print(f"restatement (port) : {r_port / 1e6:.3f} M env-steps/s ({ms_port:.2f} ms per {N}-env step)")
    print(f"port / reference   : {r_port / r_ref:.2f}x")


if __name__ == "__main__":
    main()
