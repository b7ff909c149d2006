"""Seed sweep of the env-step parity check (tests/test_env_gpu.py::test_full_size_against_cpu_oracle with the seeds, env count, terrain
and step count drawn at random): the HIP path against the CPU oracle -- masks / ids bit-exact, floats 1e-5.  Test infrastructure, run on
the GPU box:  python tools/fuzz_parity.py [cases] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from _util import FLOAT_TOL, assert_close
from isaaclab_amd.env import ManagerBasedRLEnv, load_task_cfg
from isaaclab_amd.robots import ROBOTS
from isaaclab_amd.state_feed import StateFeed
from isaaclab_amd.terrain import make_rough_terrain
from oracle.mdp_oracle import OracleEnv

TASKS = ["Isaac-Velocity-Rough-Anymal-C-v0", "Isaac-Velocity-Rough-G1-v0", "Isaac-Velocity-Flat-Anymal-C-v0", "Isaac-Cartpole-v0"]


def one_case(case_seed: int) -> str:
    rng = np.random.default_rng(case_seed)
    task = TASKS[int(rng.integers(0, len(TASKS)))]
    N = int(rng.choice([1, 2, 63, 64, 65, 127, 257, 1000, 2049, 4096, 5003]))
    steps = int(rng.integers(2, 6))
    fx = load_task_cfg(task)
    robot = ROBOTS[fx["robot"]]
    rough = "Rough" in task
    terrain = ext = None
    if rough:
        v, t, e = make_rough_terrain(int(rng.integers(2, 5)), int(rng.integers(2, 5)), tile=8.0, border=5.0, seed=int(rng.integers(0, 1000)))
        terrain, ext = (v, t), (e[0] - 1.0, e[1] - 1.0)
    snaps = int(rng.integers(2, 5))
    cpu_feed = StateFeed(robot, N, "cpu", seed=int(rng.integers(0, 10000)), num_snapshots=snaps, extent_xy=ext)
    gpu_feed = StateFeed.from_tensors(robot, [cpu_feed.snapshot(i) for i in range(snaps)], "cuda:0", cpu_feed.gravity_dir)
    env = ManagerBasedRLEnv(fx, state_feed=gpu_feed, terrain=terrain, terrain_cell=0.1 if rough else 0.0)
    env.materialize_ray_hits = rough
    D = env.plan.obs_dim
    orc = OracleEnv(fx["env"], robot.joint_names, robot.body_names, N, cpu_feed.__getitem__, cpu_feed.gravity_dir)
    gen = torch.Generator().manual_seed(int(rng.integers(0, 10000)))
    ep = torch.randint(0, env.max_episode_length, (N,), generator=gen)
    ep[::7] = env.max_episode_length - int(rng.integers(1, 3))
    env.reset()
    env.episode_length_buf = ep
    orc.episode_length_buf[:] = ep
    env._noise_u = torch.zeros(N, D, device="cuda:0")
    nreset = 0
    for _ in range(steps):
        a = torch.randn(N, env.plan.action_dim, generator=gen).clamp(-3, 3) * float(rng.choice([0.1, 1.0, 3.0]))
        u = torch.rand(N, D, generator=gen)
        env._noise_u.copy_(u)
        obs_dict, rew, term, tout, extras = env.step(a.cuda())
        orc.process_action(a)
        cpu_feed.advance()
        if rough:
            orc.ray_hits_w = env._ray_hits.cpu()
            # the fused ray path of the observation kernel against the fp64 brute force over all triangles (a slice of the envs)
            from oracle.mdp_oracle import quat_apply_yaw
            from oracle.raycast import raycast_f64

            ne, R = min(64, N), env.plan.num_rays
            local = torch.from_numpy(env.plan.ray_starts_local).unsqueeze(0).repeat(ne, 1, 1)
            starts = quat_apply_yaw(cpu_feed["root_quat_w"][:ne].repeat(1, R), local) + cpu_feed["root_pos_w"][:ne].unsqueeze(1)
            dirs = torch.tensor(env.plan.ray_direction).repeat(ne * R, 1)
            h64, _, _ = raycast_f64(terrain[0], terrain[1], starts.reshape(-1, 3).numpy(), dirs.numpy())
            assert_close(orc.ray_hits_w[:ne].reshape(-1, 3), torch.from_numpy(h64), FLOAT_TOL, "ray hits vs fp64 brute force")
        out = orc.post_physics_step(u)
        assert torch.equal(term.cpu(), out["terminated"]) and torch.equal(tout.cpu(), out["time_outs"]), "masks"
        assert torch.equal(env.reset_env_ids.cpu(), out["reset_env_ids"]), "reset ids"
        nreset += len(out["reset_env_ids"])
        assert_close(rew, out["reward"], FLOAT_TOL, "reward")
        assert_close(obs_dict["policy"], out["obs"], FLOAT_TOL, "obs")
        assert torch.equal(env.episode_length_buf.cpu(), orc.episode_length_buf), "episode length"
        for key, val in out["log"].items():
            assert abs(float(extras["log"][key]) - val) <= 1e-5 * max(1.0, abs(val)), key
    env.close()
    return f"{task} N={N} steps={steps} snapshots={snaps} resets={nreset}"


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad = 0
    for c in range(first, first + cases):
        try:
            print(f"case {c}: ok   {one_case(c)}  [{time.time() - t0:.0f} s]", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"case {c}: FAIL {e}", flush=True)
    print(f"{cases - bad} / {cases} cases agree")
    sys.exit(1 if bad else 0)
