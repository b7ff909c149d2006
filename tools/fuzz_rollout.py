"""Random sweep of the rollout: the three-launch step (imx_mlp_infer_act, imx_terminations_rewards_rollout, imx_observations) against the
six-launch split on identical states -- BIT-IDENTICAL storages and env buffers -- over random task, env count, rollout length, action
clipping, eager / graph and (velocity tasks) the env-owned full step; plus the storage's self-consistency against the torch modules.
Test infrastructure, run on the GPU box:  python tools/fuzz_rollout.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("IMX_RUNNER_QUIET", "1")
import numpy as np
import torch

from _util import Golden, assert_close
from isaaclab_amd.env import ManagerBasedRLEnv
from isaaclab_amd.robots import ROBOTS
from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper
from isaaclab_amd.state_feed import StateFeed

TASKS = ["Isaac-Velocity-Rough-Anymal-C-v0", "Isaac-Velocity-Rough-G1-v0", "Isaac-Velocity-Flat-Anymal-C-v0", "Isaac-Cartpole-v0"]
KEYS = ("observations", "actions", "actions_log_prob", "mu", "sigma", "values", "rewards", "dones")


def one_case(seed: int) -> str:
    rng = np.random.default_rng(seed)
    task = TASKS[int(rng.integers(0, len(TASKS)))]
    N = int(rng.choice([1, 16, 33, 64, 500, 1024, 2049, 2500, 4096]))
    T = int(rng.integers(2, 7))
    clip = float(rng.choice([0.0, 0.8, 2.0]))
    graph = bool(rng.integers(0, 2))
    if graph:  # (a captured rollout needs T to be a multiple of the feed's snapshot count)
        T = int(rng.choice([4, 8]))
    full = bool(rng.integers(0, 2)) and "Anymal" in task
    feed_seed, torch_seed, ep_seed = (int(rng.integers(0, 10000)) for _ in range(3))
    g = Golden(task)
    mesh = g.mesh()
    ext = None
    if mesh is not None:
        ext = (float(np.abs(mesh[0][:, 0]).max()) - 2.0, float(np.abs(mesh[0][:, 1]).max()) - 2.0)

    def build(fuse):
        torch.manual_seed(torch_seed)
        feed = StateFeed(ROBOTS[g.fixture["robot"]], N, "cuda:0", seed=feed_seed, num_snapshots=4, extent_xy=ext)
        if full:
            from bench import anydrive_like_net
            from isaaclab_amd.producers import ActuatorNetLSTM

            env = ManagerBasedRLEnv(g.fixture, state_feed=feed, terrain=mesh, noise_seed=11, own_managers=True, use_contact_sensor=True,
                                    use_articulation_update=True)
            lstm, head = anydrive_like_net("cuda:0")
            env.attach_actuator(ActuatorNetLSTM(N, 12, 80.0, 7.5, 120.0, lstm_layers=lstm, head=head, head_activation="softsign"))
        else:
            env = ManagerBasedRLEnv(g.fixture, state_feed=feed, terrain=mesh, noise_seed=11)
        venv = RslRlVecEnvWrapper(env, clip_actions=clip or None)
        runner = OnPolicyRunner(venv, dict(g.fixture["agent"], num_steps_per_env=T), log_dir=None, device="cuda:0", use_graph=graph)
        runner.fuse_launches = fuse
        runner.train_mode()
        gen = torch.Generator().manual_seed(ep_seed)
        ep = torch.randint(0, int(venv.max_episode_length), (env.num_envs,), generator=gen)
        ep[::5] = int(venv.max_episode_length) - 2
        venv.episode_length_buf = ep.cuda()
        return env, runner

    out = {}
    for fuse in (False, True):
        env, runner = build(fuse)
        if not runner._fusable():
            env.close()
            return f"{task} N={N}: not fusable (skipped)"
        for _ in range(2):
            runner.collect()
        torch.cuda.synchronize()
        stg = runner.alg.storage
        out[fuse] = {k: getattr(stg, k).clone() for k in KEYS}
        out[fuse].update(action=env._action.clone(), prev_action=env._prev_action.clone(), processed=env._processed_action.clone(),
                         cur_rew=runner._cur_reward_sum.clone(), cur_len=runner._cur_episode_length.clone(), last_obs=runner.last_obs.clone(),
                         ep_len=env.episode_length_buf.clone(), reset_ids=env.reset_env_ids.clone(), log_accum=runner._log_accum.clone())
        if fuse:  # the fused storage against the torch modules
            pol = runner.alg.policy
            with torch.no_grad():
                for t in range(T):
                    assert_close(stg.mu[t], pol.actor(stg.observations[t]), 1e-5, f"mu[{t}]")
                    assert_close(stg.values[t], pol.critic(stg.observations[t]), 1e-5, f"values[{t}]")
                    logp = torch.distributions.Normal(stg.mu[t], stg.sigma[t]).log_prob(stg.actions[t]).sum(-1, keepdim=True)
                    assert_close(stg.actions_log_prob[t], logp, 1e-4, f"log-prob[{t}]")
        env.close()
    for k in out[False]:
        assert torch.equal(out[False][k], out[True][k]), f"{k}: fused and split rollouts differ"
    return f"{task} N={N} T={T} clip={clip} graph={graph} full_step={full} dones={int(out[True]['dones'].sum())}"


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for c in range(first, first + cases):
        try:
            print(f"case {c}: ok   {one_case(c)}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"case {c}: FAIL {e}", flush=True)
    print(f"{cases - bad} / {cases} cases agree")
    sys.exit(1 if bad else 0)
