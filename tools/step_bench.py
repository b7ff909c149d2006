"""Per-kernel timings of the post-physics env step (k_action, k_term_rew, k_obs) at the bench configuration, HIP events on the
launch stream, eager back-to-back launches of ONE kernel (so launch overhead overlaps: this is the device-side duration).

    python tools/step_bench.py [--num-envs 4096 65536] [--task ...] [--graph]

--graph: additionally the three kernels of a step captured in one hipGraph and replayed (what the rollout graph pays per step).
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_env, obs_kernel_bytes_per_env  # noqa: E402
from isaaclab_amd import _lib  # noqa: E402
from isaaclab_amd._lib import check  # noqa: E402


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Isaac-Velocity-Rough-Anymal-C-v0")
    ap.add_argument("--num-envs", type=int, nargs="+", default=[4096, 65536])
    ap.add_argument("--terrain-tiles", type=int, nargs=2, default=(10, 20))
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    mesh = None
    for N in args.num_envs:
        fx, env, ntri = build_env(args.task, N, dev, 42, 4, tuple(args.terrain_tiles), mesh=mesh)
        mesh = env.terrain
        env.reset()
        act = torch.randn(N, env.plan.action_dim, device=dev).clamp_(-3, 3)
        L = env._lib
        st = _lib.current_stream(dev)

        def k_action():
            env._process_action(act)

        def k_term_rew():
            check(L.imx_terminations_rewards(env._plan_h, N, ctypes.byref(env._state()), ctypes.byref(env._bufs), 1, st))

        def k_obs():
            env._compute_observations(frame_current=True, finish_step_tail=True)

        def k_obs_no_tail():  # as bench.py's roofline measurement launches it
            env._compute_observations(frame_current=True)

        for _ in range(3):
            env.step(act)
        res = {"k_action": timeit(k_action), "k_term_rew": timeit(k_term_rew), "k_obs": timeit(k_obs), "k_obs without the step tail": timeit(k_obs_no_tail),
               "env.step (3 launches, eager)": timeit(lambda: env.step(act))}
        if args.graph:
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                env.step(act)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(env.feed.num_snapshots):
                    env.step(act)
            res["env.step (hipGraph replay, per step)"] = timeit(g.replay, 100) / env.feed.num_snapshots
        b = obs_kernel_bytes_per_env(env.plan) * N
        print(f"N={N} ({args.task}, {ntri} triangles)")
        for k, v in res.items():
            extra = f"   {b / v / 1e3:8.1f} GB/s algorithmic = {b / v / 1e3 / 8000:.3f} of 8 TB/s" if k == "k_obs" else ""
            print(f"   {k:40s} {v:8.2f} us{extra}")
        del env


if __name__ == "__main__":
    main()
