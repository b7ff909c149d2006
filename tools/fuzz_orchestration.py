"""Random sweep of the reset / interval orchestration launch (k_reset_orchestrate inside env.step() with the env's own EventManager /
CommandManager / CurriculumManager) against oracle/orchestration_oracle.py (pinned by the real-manager fixture): the fixture's cfg, scene
and state snapshots, but RANDOM draw tables, actions, episode lengths and step counts -- other reset patterns, timer phases, curriculum
moves and resampling sequences than the 48 recorded steps.  Simulator writes, trigger state, timers, levels / origins exact or 1e-5.
Test infrastructure, run on the GPU box:  python tools/fuzz_orchestration.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from _util import FLOAT_TOL, OrchGolden, assert_close
from isaaclab_amd.env import ManagerBasedRLEnv
from isaaclab_amd.events import TerrainImporterState
from oracle.orchestration_oracle import OrchestrationOracle

G = OrchGolden()


def random_draws(gen, rng):
    d = {}
    ref = G.draws(1)
    for n in G.term_names:
        d[n] = torch.rand(ref[n].shape, generator=gen)
    d["interval"] = torch.rand(ref["interval"].shape, generator=gen)
    d["command"] = torch.rand(ref["command"].shape, generator=gen)
    hi = int(G.t("draws/rand_levels").max()) + 1
    d["rand_levels"] = torch.randint(0, max(hi, 1), ref["rand_levels"].shape, generator=gen, dtype=ref["rand_levels"].dtype)
    return d


def feed_draws(env, d):
    for i, n in enumerate(G.interval_names):
        env.event_manager.get_term(n).interval_uniforms = d["interval"][i].cuda().contiguous()
    for n in G.term_names:
        env.event_manager.get_term(n).uniforms = d[n].cuda().contiguous()
    env._orch_draws["command"] = d["command"].cuda().contiguous()
    env._orch_draws["rand_levels"] = d["rand_levels"].cuda().contiguous()


def one_case(seed: int) -> str:
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    g, N, meta = G, G.N, G.meta
    ti = TerrainImporterState(g.t("terrain/origins").cuda(), g.t("terrain/levels0").cuda(), g.t("terrain/types").cuda(), meta["terrain"]["size_x"])
    env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), own_managers=True, terrain_importer=ti)
    init = g.t("interval/time_left_init") * float(rng.choice([1.0, 0.3, 0.05]))  # other timer phases than the fixture's
    for i, n in enumerate(g.interval_names):
        t = env.event_manager.get_term(n)
        t.time_left.copy_((init[i][:1].repeat(2) if t.is_global_time else init[i]).cuda())
    orc = OrchestrationOracle(g.fixture["env"], N, g.robot.num_joints, g.robot.body_names, meta["step_dt"], meta["max_episode_length_s"],
                              g.t("static/default_root_state"), g.t("static/default_joint_pos"), g.t("static/default_joint_vel"),
                              g.t("static/soft_joint_pos_limits"), g.t("static/soft_joint_vel_limits"), g.t("terrain/origins"),
                              g.t("terrain/levels0"), g.t("terrain/types"), meta["terrain"]["size_x"], init)
    cpu_feed = g.feed("cpu")
    ev, ct = env.event_manager, env.command_term

    def state():
        return {k: cpu_feed[k] for k in ("root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w")}

    def check(tag):
        torch.cuda.synchronize()
        for k, v in env.sim_writes.items():
            assert_close(v, orc.sim_writes[k], FLOAT_TOL, f"{tag} sim_writes[{k}]")
        assert torch.equal(ti.terrain_levels.cpu(), orc.levels), f"{tag} terrain levels"
        assert torch.equal(ti.env_origins.cpu(), orc.env_origins), f"{tag} env origins"
        c = orc.cmd
        for k, a, b in (("command", ct.vel_command_b, c.vel_command_b), ("time_left", ct.time_left, c.time_left), ("heading", ct.heading_target, c.heading_target),
                        ("error_vel_xy", ct.metrics["error_vel_xy"], c.metrics["error_vel_xy"]), ("error_vel_yaw", ct.metrics["error_vel_yaw"], c.metrics["error_vel_yaw"])):
            assert_close(a, b, FLOAT_TOL, f"{tag} command {k}")
        assert torch.equal(ct.command_counter.cpu(), c.command_counter) and torch.equal(ct.is_standing_env.cpu(), c.is_standing_env), f"{tag} command flags"
        step = int(env._counters[2])
        tl = torch.stack([(t.time_left[(step + 1) & 1].expand(N) if t.is_global_time else t.time_left) for t in (ev.get_term(n) for n in g.interval_names)])
        assert_close(tl, torch.stack([orc.time_left[n].expand(N) for n in g.interval_names]), 1e-6, f"{tag} interval timers")
        assert torch.equal(torch.stack([ev.get_term(n).last_triggered_step for n in g.reset_names]).cpu(), torch.stack([orc.last_triggered[n] for n in g.reset_names])), f"{tag} last triggered"
        assert torch.equal(torch.stack([ev.get_term(n).triggered_once for n in g.reset_names]).cpu(), torch.stack([orc.triggered_once[n] for n in g.reset_names])), f"{tag} triggered once"

    d = random_draws(gen, rng)
    feed_draws(env, d)
    env.reset()
    orc.cmd._draw[:] = 0
    orc.reset_idx(torch.arange(N), state(), 0, d, d["command"], d["rand_levels"])
    check("reset")
    ep = torch.randint(0, env.max_episode_length, (N,), generator=gen)
    ep[:: int(rng.choice([2, 3, 7]))] = env.max_episode_length - int(rng.integers(1, 4))
    env.episode_length_buf = ep
    steps = int(rng.integers(5, 40))
    resets = pushes = 0
    scale = float(rng.choice([0.2, 1.0, 3.0]))
    for s in range(steps):
        d = random_draws(gen, rng)
        feed_draws(env, d)
        env.step((torch.randn(N, env.plan.action_dim, generator=gen) * scale).clamp(-3, 3).cuda())
        cpu_feed.advance()
        ids = env.reset_env_ids.cpu()
        orc.cmd._draw[:] = 0
        if len(ids):
            orc.reset_idx(ids, state(), s + 1, d, d["command"], d["rand_levels"])
            resets += len(ids)
        fired = orc.step_tail(state(), d, d["interval"], d["command"])
        pushes += sum(len(v) for v in fired.values())
        check(f"step {s}")
    env.close()
    return f"steps={steps} resets={resets} pushes={pushes} mean level {float(orc.levels.float().mean()):.2f}"


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for c in range(first, first + cases):
        try:
            print(f"case {c}: ok   {one_case(c)}", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"case {c}: FAIL {str(e)[:400]}", flush=True)
    print(f"{cases - bad} / {cases} cases agree")
    sys.exit(1 if bad else 0)
