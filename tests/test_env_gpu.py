"""GPU parity: the HIP path (through the C ABI) against the golden fixtures produced by the real reference, and
against the CPU oracle at the benchmark size.  Masks / indices / counters bit-exact, floats within 1e-5."""

import numpy as np
import pytest
import torch

from _util import FLOAT_TOL, TASKS, Golden, assert_close

pytestmark = pytest.mark.gpu


def make_env(g: Golden, **kw):
    from isaaclab_amd.env import ManagerBasedRLEnv

    feed = g.feed("cuda:0")
    mesh = g.mesh()
    return ManagerBasedRLEnv(g.fixture, state_feed=feed, terrain=mesh, **kw)


@pytest.mark.parametrize("tail", ["deferred", "in_kernel"])
@pytest.mark.parametrize("task", TASKS)
def test_env_step_matches_reference_goldens(task, tail):
    """``tail``: the end of the step (ordered reset ids, reset count, Episode_* log) finished by an extra workgroup of the observation
    launch (what env.step() does) or inside imx_terminations_rewards by its last-arriving workgroup (the stand-alone C-ABI call)."""
    g = Golden(task)
    env = make_env(g)
    env.defer_step_tail = tail == "deferred"
    env.materialize_ray_hits = True
    N, D = g.N, g.meta["obs_dim"]
    env._noise_u = torch.zeros(N, D, device="cuda:0")
    env._noise_u.copy_(g.t("reset/noise_u"))
    obs_dict, extras = env.reset()
    assert obs_dict["policy"].shape == (N, D)
    assert_close(obs_dict["policy"], g.t("reset/obs"), FLOAT_TOL, "reset obs")
    if g.mesh() is not None:
        assert_close(env._ray_hits, g.t("reset/ray_hits_w"), FLOAT_TOL, "ray hits (fp64 brute-force oracle)")
    env.episode_length_buf = g.t("reset/episode_length_buf")
    names_r, names_t = g.meta["reward_terms"], g.meta["termination_terms"]
    for k in range(g.steps):
        tag = f"step{k}"
        env._noise_u.copy_(g.t(f"{tag}/noise_u"))
        obs_dict, rew, terminated, time_outs, extras = env.step(g.t(f"{tag}/action").cuda())
        torch.cuda.synchronize()
        # -- bit-exact: masks, indices, counters
        assert torch.equal(terminated.cpu(), g.t(f"{tag}/terminated")), "terminated"
        assert torch.equal(time_outs.cpu(), g.t(f"{tag}/time_outs")), "time_outs"
        assert torch.equal(env.reset_buf.cpu(), g.t(f"{tag}/reset_buf")), "reset_buf"
        assert torch.equal(env.reset_env_ids.cpu(), g.t(f"{tag}/reset_env_ids")), "reset_env_ids"
        for name in names_t:
            assert torch.equal(env.termination_manager.get_term(name).cpu(), g.t(f"{tag}/term_dones/{name}")), name
        assert torch.equal(env.episode_length_buf.cpu(), g.t(f"{tag}/episode_length_buf")), "episode_length_buf"
        # -- floats
        assert_close(torch.cat([env.action_manager.get_term(n).processed_actions for n in env.action_manager.active_terms], dim=1),
                     g.t(f"{tag}/processed_actions"), FLOAT_TOL, "processed_actions")
        assert_close(rew, g.t(f"{tag}/reward"), FLOAT_TOL, "reward")
        assert_close(env.reward_manager._step_reward, g.t(f"{tag}/step_reward"), FLOAT_TOL, "step_reward")
        for name in names_r:
            assert_close(env.reward_manager._episode_sums[name], g.t(f"{tag}/episode_sums/{name}"), FLOAT_TOL, name)
        assert_close(env.action_manager.action, g.t(f"{tag}/action_after_reset"), 0.0, "action after reset")
        assert_close(env.action_manager.prev_action, g.t(f"{tag}/prev_action_after_reset"), 0.0, "prev_action")
        assert_close(obs_dict["policy"], g.t(f"{tag}/obs"), FLOAT_TOL, "obs")
        if g.mesh() is not None:
            assert_close(env._ray_hits, g.t(f"{tag}/ray_hits_w"), FLOAT_TOL, "ray hits")
        ref_log = g.log(k)
        for key, v in ref_log.items():
            got = float(extras["log"][key])
            assert abs(got - v) <= FLOAT_TOL * max(1.0, abs(v)), (key, got, v)
    env.close()


def test_group_shapes_match_reference():
    """The other shapes ObservationManager hands out (observation_manager.py:320-335) -- a dict-of-terms group, un-flattened (N, H, d)
    history terms, a concatenated (N, H, sum d) group -- are views of the row the one fused launch fills; fixture from the real manager,
    5 steps with resets (history windows of reset envs refill with their first value)."""
    from _util import SHAPES, assert_groups_close

    g = Golden(SHAPES)
    env = make_env(g)
    om = env.observation_manager
    assert om.group_obs_concatenate == g.meta["obs_group_concatenate"]
    assert {k: [list(d) for d in v] for k, v in om.group_obs_term_dim.items()} == g.meta["obs_group_term_shapes"]
    shapes = {k: ([list(d) for d in v] if isinstance(v, list) else list(v)) for k, v in om.group_obs_dim.items()}
    assert shapes == g.meta["obs_group_shapes"]  # incl. the reference's element-wise sum for the stacked group: (6, 9)
    assert env.single_observation_space["terms"]["base_lin_vel"].shape == (2, 3) and env.observation_space["terms"]["actions"].shape == (g.N, 36)
    env._noise_u = g.t("reset/noise_u").cuda()
    obs_dict, _ = env.reset()
    assert_groups_close(obs_dict, g, "reset", FLOAT_TOL)
    env.episode_length_buf = g.t("reset/episode_length_buf")
    for k in range(g.steps):
        tag = f"step{k}"
        env._noise_u.copy_(g.t(f"{tag}/noise_u"))
        obs_dict, rew, terminated, time_outs, extras = env.step(g.t(f"{tag}/action").cuda())
        assert torch.equal(env.reset_env_ids.cpu(), g.t(f"{tag}/reset_env_ids"))
        assert_close(rew, g.t(f"{tag}/reward"), FLOAT_TOL, "reward")
        assert_groups_close(obs_dict, g, tag, FLOAT_TOL)
    assert obs_dict["stack"].shape == (g.N, 2, 9) and obs_dict["terms"]["joint_vel"].shape == (g.N, 2, 12)
    with pytest.raises(ValueError):
        om.compute_group("nope")
    env.close()


def test_wrapper_surface_and_time_outs():
    from isaaclab_amd.rsl_rl import RslRlVecEnvWrapper

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    env = RslRlVecEnvWrapper(make_env(g), clip_actions=1.0)
    assert (env.num_envs, env.num_obs, env.num_actions, env.num_privileged_obs) == (64, 48, 12, 0)
    obs, extras = env.get_observations()
    assert obs.shape == (64, 48) and "observations" in extras
    a = torch.randn(64, 12, device="cuda:0") * 3
    obs, rew, dones, extras = env.step(a)
    assert dones.dtype == torch.long and rew.shape == (64,)
    assert "time_outs" in extras and extras["time_outs"].dtype == torch.bool  # infinite horizon (vecenv_wrapper.py:184)
    assert float(env.unwrapped.action_manager.action.abs().max()) <= 1.0  # clip fused into imx_action_process
    with pytest.raises(ValueError):
        env.step(torch.zeros(64, 5, device="cuda:0"))  # action_manager.py:328-329
    env.episode_length_buf = torch.full((64,), 7, device="cuda:0")
    assert int(env.unwrapped.episode_length_buf[3]) == 7


@pytest.mark.parametrize("task,N", [("Isaac-Velocity-Rough-Anymal-C-v0", 4096), ("Isaac-Velocity-Rough-G1-v0", 4096),
                                    ("Isaac-Velocity-Flat-Anymal-C-v0", 100_003),
                                    # beyond 8192 envs: the observation kernel's env-major wave order, 32-env groups in the step kernel
                                    ("Isaac-Velocity-Rough-Anymal-C-v0", 12_001),
                                    ("Isaac-Velocity-Rough-Anymal-C-v0", 20_011),  # ... and 64-env groups beyond 16 384
                                    # ragged / tiny batches: partial waves, a single env, one env past a 64-env block
                                    ("Isaac-Velocity-Rough-Anymal-C-v0", 1), ("Isaac-Velocity-Rough-Anymal-C-v0", 63),
                                    ("Isaac-Velocity-Flat-Anymal-C-v0", 65), ("Isaac-Cartpole-v0", 3)])
def test_full_size_against_cpu_oracle(task, N):
    """BASELINE.json sizes: same seeded synthetic feed through the HIP path and the CPU oracle."""
    from isaaclab_amd.env import ManagerBasedRLEnv, load_task_cfg
    from isaaclab_amd.robots import ROBOTS
    from isaaclab_amd.state_feed import StateFeed
    from isaaclab_amd.terrain import make_rough_terrain
    from oracle.mdp_oracle import OracleEnv

    fx = load_task_cfg(task)
    robot = ROBOTS[fx["robot"]]
    rough = "Rough" in task
    terrain = ext = None
    if rough:
        v, t, e = make_rough_terrain(4, 6, tile=8.0, border=5.0, seed=3)
        terrain, ext = (v, t), (e[0] - 1.0, e[1] - 1.0)
    cpu_feed = StateFeed(robot, N, "cpu", seed=9, num_snapshots=3, extent_xy=ext)
    gpu_feed = StateFeed.from_tensors(robot, [cpu_feed.snapshot(i) for i in range(3)], "cuda:0", cpu_feed.gravity_dir)
    env = ManagerBasedRLEnv(fx, state_feed=gpu_feed, terrain=terrain, terrain_cell=0.1 if rough else 0.0)
    env.materialize_ray_hits = rough
    D = env.plan.obs_dim
    orc = OracleEnv(fx["env"], robot.joint_names, robot.body_names, N, cpu_feed.__getitem__, cpu_feed.gravity_dir)
    gen = torch.Generator().manual_seed(5)
    ep = torch.randint(0, env.max_episode_length, (N,), generator=gen)
    ep[::97] = env.max_episode_length - 1
    acts = [torch.randn(N, env.plan.action_dim, generator=gen).clamp(-3, 3) for _ in range(2)]
    us = [torch.rand(N, D, generator=gen) for _ in range(2)]
    env.reset()
    env.episode_length_buf = ep
    orc.episode_length_buf[:] = ep
    env._noise_u = torch.zeros(N, D, device="cuda:0")
    recorded = []
    for a, u in zip(acts, us):
        env._noise_u.copy_(u)
        obs_dict, rew, term, tout, extras = env.step(a.cuda())
        recorded.append((obs_dict["policy"].clone(), rew.clone(), env.reset_buf.clone()))
        orc.process_action(a)
        cpu_feed.advance()
        if rough:  # the oracle does not ray-cast: give it the HIP hits (checked separately against brute force)
            orc.ray_hits_w = env._ray_hits.cpu()
            # ... and a slice of them against the fp64 brute-force oracle (misses = +inf are legitimate at mesh seams)
            from oracle.mdp_oracle import quat_apply_yaw
            from oracle.raycast import raycast_f64

            ne, R = min(256, N), env.plan.num_rays  # 48 k rays: lattice, FLAT and edge cells of the mixed terrain
            local = torch.from_numpy(env.plan.ray_starts_local).unsqueeze(0).repeat(ne, 1, 1)
            starts = quat_apply_yaw(cpu_feed["root_quat_w"][:ne].repeat(1, R), local) + cpu_feed["root_pos_w"][:ne].unsqueeze(1)
            dirs = torch.tensor(env.plan.ray_direction).repeat(ne * R, 1)
            h64, _, _ = raycast_f64(terrain[0], terrain[1], starts.reshape(-1, 3).numpy(), dirs.numpy())
            assert_close(orc.ray_hits_w[:ne].reshape(-1, 3), torch.from_numpy(h64), FLOAT_TOL, "ray hits vs brute force")
            assert torch.isfinite(orc.ray_hits_w).float().mean() > 0.99
        out = orc.post_physics_step(u)
        assert torch.equal(term.cpu(), out["terminated"]) and torch.equal(tout.cpu(), out["time_outs"])
        assert torch.equal(env.reset_env_ids.cpu(), out["reset_env_ids"])
        assert len(out["reset_env_ids"]) > 0 or (N < 97 and len(recorded) > 1)  # small batches: only env 0 is primed to time out, on the first step
        assert_close(rew, out["reward"], FLOAT_TOL, "reward")
        assert_close(obs_dict["policy"], out["obs"], FLOAT_TOL, "obs")
        assert torch.equal(env.episode_length_buf.cpu(), orc.episode_length_buf)
        for key, v in out["log"].items():
            assert abs(float(extras["log"][key]) - v) <= 1e-5 * max(1.0, abs(v)), key
    # size-independent property: a second env fed the same tensors is bit-identical (determinism, cf.
    # source/isaaclab_tasks/test/test_environment_determinism.py)
    feed2 = StateFeed.from_tensors(robot, [cpu_feed.snapshot(i) for i in range(3)], "cuda:0", cpu_feed.gravity_dir)
    env2 = ManagerBasedRLEnv(fx, state_feed=feed2, terrain=env.terrain)
    env2.reset()
    env2.episode_length_buf = ep
    env2._noise_u = torch.zeros(N, D, device="cuda:0")
    for (a, u), (o, r, rb) in zip(zip(acts, us), recorded):
        env2._noise_u.copy_(u)
        obs_dict, rew, _, _, _ = env2.step(a.cuda())
        assert torch.equal(obs_dict["policy"], o) and torch.equal(rew, r) and torch.equal(env2.reset_buf, rb)
    env.close()
    env2.close()


def test_height_scanner_rays_along_cell_boundaries():
    """Every ray of every env within a few 1e-5 m of a grid line of the terrain's cell grid (sensor on a grid node +- 2e-5 .. 5e-5 m,
    yaw a multiple of 90 degrees, ray pattern pitch = cell size): the paths for rays within tau of a cell boundary -- neighbour cells,
    reference lists, or the builder's proof that the cell's own record is complete there -- against the fp64 brute force over ALL
    triangles, on box, stair and height-field tiles."""
    from isaaclab_amd.env import ManagerBasedRLEnv, load_task_cfg
    from isaaclab_amd.robots import ROBOTS
    from isaaclab_amd.state_feed import StateFeed
    from isaaclab_amd.terrain import make_rough_terrain
    from oracle.mdp_oracle import quat_apply_yaw
    from oracle.raycast import raycast_f64

    task, N = "Isaac-Velocity-Rough-Anymal-C-v0", 512
    fx = load_task_cfg(task)
    robot = ROBOTS[fx["robot"]]
    v, t, e = make_rough_terrain(4, 6, tile=8.0, border=5.0, seed=3)
    feed = StateFeed(robot, N, "cpu", seed=11, num_snapshots=2, extent_xy=(e[0] - 1.0, e[1] - 1.0))
    gen = torch.Generator().manual_seed(3)
    pos, quat = feed._stack["root_pos_w"], feed._stack["root_quat_w"]
    node = torch.stack([torch.randint(-140, 141, (N,), generator=gen), torch.randint(-220, 221, (N,), generator=gen)], 1).float() * 0.1
    jitter = torch.tensor([-5e-5, -2e-5, 2e-5, 5e-5])[torch.randint(0, 4, (N, 2), generator=gen)]
    jitter[::7] = torch.tensor([3e-4, -3e-4])  # and some clear of the tau band, for contrast
    pos[:, :, :2] = (node + jitter).unsqueeze(0)
    yaw = torch.randint(0, 4, (N,), generator=gen).float() * (torch.pi / 2)
    quat[:] = torch.stack([torch.cos(yaw / 2), torch.zeros(N), torch.zeros(N), torch.sin(yaw / 2)], 1).unsqueeze(0)
    gpu_feed = StateFeed.from_tensors(robot, [feed.snapshot(i) for i in range(2)], "cuda:0", feed.gravity_dir)
    env = ManagerBasedRLEnv(fx, state_feed=gpu_feed, terrain=(v, t), terrain_cell=0.1)
    env.materialize_ray_hits = True
    env.plan.enable_corruption = False
    obs_dict, _ = env.reset()
    R = env.plan.num_rays
    local = torch.from_numpy(env.plan.ray_starts_local).unsqueeze(0).repeat(N, 1, 1)
    starts = quat_apply_yaw(feed["root_quat_w"].repeat(1, R), local) + feed["root_pos_w"].unsqueeze(1)
    # the premise: (almost) every ray sits within tau = 1e-3 cells of a grid line
    g = (starts[..., :2].reshape(-1, 2).double() - torch.tensor([float(v[:, 0].min()), float(v[:, 1].min())]).double()) / 0.1
    near = ((g - g.round()).abs() < 1e-3).any(1).float().mean()
    assert near > 0.8, float(near)
    dirs = torch.tensor(env.plan.ray_direction).repeat(N * R, 1)
    h64, _, _ = raycast_f64(v, t, starts.reshape(-1, 3).numpy(), dirs.numpy())
    got = env._ray_hits.cpu().reshape(-1, 3)
    assert torch.isfinite(got).all(1).float().mean() > 0.99
    assert_close(got, torch.from_numpy(h64), FLOAT_TOL, "boundary rays vs fp64 brute force")
    # and through to the observation columns: height_scan = sensor z - hit z - offset, clipped
    scan = obs_dict["policy"][:, -R:].cpu()
    want = (feed["root_pos_w"][:, 2:3] - torch.from_numpy(h64).view(N, R, 3)[..., 2] - 0.5).clip(-1.0, 1.0)
    assert_close(scan, want.float(), FLOAT_TOL, "height_scan columns")
    env.close()


def test_in_kernel_noise_is_uniform_and_bounded():
    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    env = make_env(g)
    env.reset()
    a = torch.zeros(64, 12, device="cuda:0")
    o1 = env.step(a)[0]["policy"].clone()
    env.plan.enable_corruption = False
    env.feed.seek(env.feed.index)
    clean = env._compute_observations().clone()
    d = (o1 - clean)
    # base_lin_vel columns: U(-0.1, 0.1); command columns: no noise
    assert float(d[:, 0:3].abs().max()) <= 0.1 + 1e-6 and float(d[:, 0:3].abs().max()) > 0.05
    assert float(d[:, 9:12].abs().max()) == 0.0
    assert abs(float(d[:, 24:36].mean())) < 0.2  # joint_vel noise U(-1.5,1.5), 768 samples
    # the gaussian generator (Box-Muller on the counter-based uniforms): actions = N(-0.2, 0.3) ("abs"), base_lin_vel += N(0.01, 0.1),
    # base_ang_vel += 0.05 (constant_noise draws nothing); 64 envs x 12 / 3 columns x 3 steps of fresh samples
    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0-noise")
    env = make_env(g)
    env.reset()
    acts, dl, da = [], [], []
    for _ in range(6):
        o = env.step(a)[0]["policy"].clone()
        env.plan.enable_corruption = False
        env.feed.seek(env.feed.index)
        clean = env._compute_observations().clone()
        env.plan.enable_corruption = True
        acts.append(o[:, 36:48]); dl.append(o[:, 0:3] - clean[:, 0:3]); da.append(o[:, 3:6] - clean[:, 3:6])
    z = (torch.cat(acts) + 0.2) / 0.3
    assert abs(float(z.mean())) < 0.06 and abs(float(z.std()) - 1.0) < 0.05 and float(z.abs().max()) > 2.5  # 4608 samples
    assert not torch.equal(acts[0], acts[1])  # fresh every step
    zl = (torch.cat(dl) - 0.01) / 0.1
    assert abs(float(zl.mean())) < 0.1 and abs(float(zl.std()) - 1.0) < 0.1
    assert float((torch.cat(da) - 0.05).abs().max()) < 1e-6


def test_env_owned_command_term_and_contact_sensor():
    """SURVEY 8f row 1 wired into the env: commands are resampled by imx_velocity_command (reset envs + timer) and the
    contact sensor state is advanced by imx_contact_sensor_update from the feed's per-step forces."""
    from isaaclab_amd.env import ManagerBasedRLEnv

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), use_command_term=True, use_contact_sensor=True)
    obs, _ = env.reset()
    cmd0 = env.command_manager.get_command("base_velocity").clone()
    assert float(cmd0.abs().max()) <= 1.0 and float(cmd0.abs().sum()) > 0  # resampled at reset inside the cfg ranges
    assert torch.equal(obs["policy"][:, 9:12], cmd0)  # velocity_commands columns (no noise on this term)
    assert int(env.command_term.command_counter.min()) == 1
    a = torch.zeros(64, 12, device="cuda:0")
    env.episode_length_buf = torch.full((64,), 998)
    for k in range(3):
        counter_before = env.command_term.command_counter.clone()
        obs, rew, term, tout, _ = env.step(a)
        c = env.command_manager.get_command("base_velocity")
        assert torch.equal(obs["policy"][:, 9:12], c)
        stand = env.command_term.is_standing_env
        assert bool((c[stand] == 0).all()) and float(c.abs().max()) <= 1.0 + 1e-6
        if k == 1:  # 998 + 2 steps = the 1000-step limit: every env times out and gets a fresh command (counter -> 1)
            assert bool(tout.all())
            assert torch.equal(env.command_term.command_counter, torch.ones(64, dtype=torch.long, device="cuda:0"))
        else:
            assert not bool(tout.any()) and torch.equal(env.command_term.command_counter[~term], counter_before[~term])
    cs = env.contact_sensor
    assert torch.equal(cs.data.net_forces_w_history[:, 0], env.feed["net_forces_w_history"][:, 0])
    # every env timed out at k == 1: _reset_idx -> scene.reset(env_ids) -> ContactSensor.reset zeroed the sensor clocks (contact_sensor.py:143-165),
    # one more step since
    assert float(cs._timestamp.max()) == pytest.approx(1 * env.step_dt, rel=1e-5)
    assert torch.isfinite(rew).all()
    env.close()


def test_env_owned_reset_events():
    """SURVEY 8f row 2 wired into the env: the step kernel's reset mask drives imx_reset_events; what the reference's
    EventManager would write to the simulator lands in env.sim_writes for exactly the envs that reset."""
    from isaaclab_amd.env import ManagerBasedRLEnv

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    events = {"reset_base": {"func": "isaaclab.envs.mdp.events:reset_root_state_uniform", "mode": "reset",
                             "params": {"pose_range": {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14)}, "velocity_range": {}}},
              "reset_robot_joints": {"func": "isaaclab.envs.mdp.events:reset_joints_by_scale", "mode": "reset",
                                     "params": {"position_range": (0.5, 1.5), "velocity_range": (0.0, 0.0)}}}
    env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), events_cfg=events)
    assert env.event_manager.active_terms == {"reset": ["reset_base", "reset_robot_joints"]}
    env.reset()  # ManagerBasedEnv.reset -> _reset_idx(every env): the reset events run on all of them (manager_based_env.py:264-315)
    assert bool((env.sim_writes["root_pose"][:, 3:7].norm(dim=-1) > 0.99).all())
    for v in env.sim_writes.values():
        v.zero_()
    a = torch.zeros(64, 12, device="cuda:0")
    ever = torch.zeros(64, dtype=torch.bool, device="cuda:0")
    for _ in range(4):
        before = {k: v.clone() for k, v in env.sim_writes.items()}
        _, _, term, tout, _ = env.step(a)
        done = term | tout
        ever |= done
        for k, v in env.sim_writes.items():
            assert torch.equal(v[~done], before[k][~done])  # untouched rows
        d = env.sim_writes["root_pose"][done, :3] - env.feed["env_origins"][done] - env.default_root_state[done, :3]
        if done.any():
            assert float(d[:, :2].abs().max()) <= 0.5 + 1e-6 and float(d[:, 2].abs().max()) <= 1e-6
            jp, dj = env.sim_writes["joint_pos"][done], env.feed["default_joint_pos"][done]
            lim = env.feed["soft_joint_pos_limits"][done]
            expect_lo = torch.minimum(dj * 0.5, dj * 1.5).clamp(lim[..., 0], lim[..., 1])
            expect_hi = torch.maximum(dj * 0.5, dj * 1.5).clamp(lim[..., 0], lim[..., 1])
            assert bool((jp >= expect_lo - 1e-6).all()) and bool((jp <= expect_hi + 1e-6).all())
    assert bool(ever.any()) and not bool(ever.all())
    assert bool((env.sim_writes["root_pose"][~ever] == 0).all())
    env.close()


def _orch_env(g, **kw):
    from isaaclab_amd.env import ManagerBasedRLEnv
    from isaaclab_amd.events import TerrainImporterState

    ti = TerrainImporterState(g.t("terrain/origins").cuda(), g.t("terrain/levels0").cuda(), g.t("terrain/types").cuda(), g.meta["terrain"]["size_x"])
    env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), own_managers=True, terrain_importer=ti, **kw)
    init = g.t("interval/time_left_init").cuda()
    for i, n in enumerate(g.interval_names):  # EventManager._prepare_terms drew these: take the fixture's
        t = env.event_manager.get_term(n)
        t.time_left.copy_(init[i][:1].repeat(2) if t.is_global_time else init[i])
    return env


def _orch_feed_draws(env, g, slot):
    d = g.draws(slot)
    for i, n in enumerate(g.interval_names):
        env.event_manager.get_term(n).interval_uniforms = d["interval"][i].cuda().contiguous()
    for n in g.term_names:
        env.event_manager.get_term(n).uniforms = d[n].cuda().contiguous()
    env._orch_draws["command"] = d["command"].cuda().contiguous()
    env._orch_draws["rand_levels"] = d["rand_levels"].cuda().contiguous()


def test_reset_and_interval_orchestration_matches_the_real_managers():
    """env.step() with the env's OWN EventManager / CommandManager / CurriculumManager (``own_managers=True``: one imx_reset_orchestrate
    launch between the step kernel and the observations) against 48 steps of the REAL ``ManagerBasedRLEnv._reset_idx`` + EventManager
    (``min_step_count_between_reset``, per-env and global interval timers) + CommandManager + CurriculumManager + terrain_levels_vel,
    fed the reference's recorded draws: what lands in the simulator writes, trigger state and timers bit-exact, commands / metrics /
    observations / rewards / log entries (incl. ``Metrics/*`` and ``Curriculum/*``) within 1e-5, terrain levels and origins exact."""
    from _util import OrchGolden

    g = OrchGolden()
    N = g.N
    env = _orch_env(g)
    assert env.event_manager.active_terms == g.meta["event_terms"]
    assert env.curriculum_manager.active_terms == ["terrain_levels"] and env.event_manager.skipped_terms == []
    ev, ct, ti = env.event_manager, env.command_term, env.terrain_importer

    def check(tag, extras):
        torch.cuda.synchronize()
        for k, v in env.sim_writes.items():
            assert_close(v, g.t(f"{tag}/sim_writes/{k}"), FLOAT_TOL, f"{tag} sim_writes[{k}]")
        assert torch.equal(ti.terrain_levels.cpu(), g.t(f"{tag}/terrain_levels")), f"{tag} terrain levels"
        assert torch.equal(ti.env_origins.cpu(), g.t(f"{tag}/env_origins")), f"{tag} env origins"
        for k, a in (("command", ct.vel_command_b), ("command_time_left", ct.time_left), ("heading_target", ct.heading_target),
                     ("metric_error_vel_xy", ct.metrics["error_vel_xy"]), ("metric_error_vel_yaw", ct.metrics["error_vel_yaw"])):
            assert_close(a, g.t(f"{tag}/{k}"), FLOAT_TOL, f"{tag} {k}")
        assert torch.equal(ct.command_counter.cpu(), g.t(f"{tag}/command_counter")), f"{tag} command counter"
        assert torch.equal(ct.is_standing_env.cpu(), g.t(f"{tag}/is_standing_env")) and torch.equal(ct.is_heading_env.cpu(), g.t(f"{tag}/is_heading_env"))
        step = int(env._counters[2])
        tl = torch.stack([(t.time_left[(step + 1) & 1].expand(N) if t.is_global_time else t.time_left) for t in (ev.get_term(n) for n in g.interval_names)])
        assert_close(tl, g.t(f"{tag}/interval_time_left"), 1e-6, f"{tag} interval timers")
        assert torch.equal(torch.stack([ev.get_term(n).last_triggered_step for n in g.reset_names]).cpu(), g.t(f"{tag}/reset_last_triggered_step")), tag
        assert torch.equal(torch.stack([ev.get_term(n).triggered_once for n in g.reset_names]).cpu(), g.t(f"{tag}/reset_triggered_once")), tag
        for key, v in g.log(tag).items():
            got = float(extras["log"][key])
            assert abs(got - v) <= FLOAT_TOL * max(1.0, abs(v)), (tag, key, got, v)

    _orch_feed_draws(env, g, 0)
    obs, extras = env.reset()
    assert_close(obs["policy"], g.t("reset/obs"), FLOAT_TOL, "reset obs")
    check("reset", extras)
    env.episode_length_buf = g.t("reset/episode_length_buf")
    seen = set()
    for k in range(g.steps):
        tag = f"step{k}"
        _orch_feed_draws(env, g, k + 1)
        obs, rew, terminated, time_outs, extras = env.step(g.t(f"{tag}/action").cuda())
        assert torch.equal(terminated.cpu(), g.t(f"{tag}/terminated")) and torch.equal(time_outs.cpu(), g.t(f"{tag}/time_outs"))
        assert torch.equal(env.reset_env_ids.cpu(), g.t(f"{tag}/reset_env_ids"))
        assert torch.equal(env.episode_length_buf.cpu(), g.t(f"{tag}/episode_length_buf"))
        assert_close(rew, g.t(f"{tag}/reward"), FLOAT_TOL, f"{tag} reward")
        assert_close(obs["policy"], g.t(f"{tag}/obs"), FLOAT_TOL, f"{tag} obs")
        check(tag, extras)
        seen |= set(g.log(tag))
    assert {"Metrics/base_velocity/error_vel_xy", "Metrics/base_velocity/error_vel_yaw", "Curriculum/terrain_levels"} <= seen
    env.close()


def test_orchestration_in_kernel_draws_and_graph_capture():
    """Performance mode of the same launch (no recorded draws: the counter-based generator): samples inside the cfg ranges, interval
    timers re-armed inside [lo, hi], a captured rollout of env steps replays with fresh draws, `seed()` makes runs repeat."""
    from _util import OrchGolden

    g = OrchGolden()
    out = []
    for trial in range(2):
        env = _orch_env(g, seed=7)
        env.reset()
        env.episode_length_buf = g.t("reset/episode_length_buf")
        a = torch.zeros(g.N, 12, device="cuda:0")
        S = env.feed.num_snapshots
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            env.step(a)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(7):
                env.step(a)
        pushes = []
        for _ in range(6):
            gr.replay()
            pushes.append(env.sim_writes["root_vel"].clone())
        torch.cuda.synchronize()
        assert not torch.equal(pushes[-1], pushes[-2])  # fresh draws on every replay (keyed by the device step counter)
        t = env.event_manager.get_term("push_robot")
        lo, hi = t.interval_range_s
        assert float(t.time_left.max()) <= hi + 1e-6 and float(t.time_left.min()) > -env.step_dt
        c = env.command_term.vel_command_b
        assert float(c.abs().max()) <= 1.0 + 1e-6
        jp = env.sim_writes["joint_pos"]
        lim = env.feed["soft_joint_pos_limits"]
        assert bool((jp >= lim[..., 0] - 1e-6).all()) and bool((jp <= lim[..., 1] + 1e-6).all())
        assert int(env.terrain_importer.terrain_levels.min()) >= 0 and int(env.terrain_importer.terrain_levels.max()) < g.meta["terrain"]["rows"]
        out.append((pushes[-1].clone(), env.terrain_importer.terrain_levels.clone(), c.clone()))
        env.close()
    assert all(torch.equal(x, y) for x, y in zip(out[0], out[1])), "same seed, same rollout"


def test_runner_logs_metrics_and_curriculum_of_an_env_with_its_own_managers(tmp_path):
    """OnPolicyRunner.learn over an env that owns its Event / Command / Curriculum managers (three-launch rollout in a hipGraph, the
    orchestration launch inside it): the per-iteration means of ``Metrics/<command>/*`` and ``Curriculum/terrain_levels`` reach the
    scalar log next to ``Episode_Reward/*`` (manager_based_rl_env.py:365-389 -> upstream runner's ep_infos), all summed on the device."""
    import json as _json

    from _util import OrchGolden
    from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

    g = OrchGolden()
    env = _orch_env(g, seed=3)
    venv = RslRlVecEnvWrapper(env)
    runner = OnPolicyRunner(venv, dict(g.fixture["agent"], num_steps_per_env=49), log_dir=str(tmp_path), device="cuda:0", use_graph=True)  # (49 snapshots in the recorded feed: the graph bakes one pass over them)
    assert runner._fusable()
    venv.episode_length_buf = g.t("reset/episode_length_buf")
    import os as _os

    _os.environ["IMX_RUNNER_QUIET"] = "1"
    try:
        runner.learn(3)
    finally:
        _os.environ.pop("IMX_RUNNER_QUIET", None)
    last = runner.writer.last
    for key in ("Metrics/base_velocity/error_vel_xy", "Metrics/base_velocity/error_vel_yaw", "Curriculum/terrain_levels",
                "Episode_Reward/track_lin_vel_xy_exp", "Episode_Termination/time_out", "Perf/total_fps"):
        assert key in last and last[key] == last[key], key
    R = g.meta["terrain"]["rows"]
    assert 0.0 <= last["Curriculum/terrain_levels"] <= R - 1 and last["Metrics/base_velocity/error_vel_xy"] >= 0.0
    rows = [_json.loads(ln) for ln in open(_os.path.join(str(tmp_path), "scalars.jsonl"))]
    assert sum(r["tag"] == "Curriculum/terrain_levels" for r in rows) == 3
    env.close()


def test_orchestration_subsets_and_guards():
    """Parts of the orchestration on their own (a global-time push only; events without a command term), the events given as the cfg's own
    section, unknown terms refused by name, the non-deferred step tail refused for an env with its own managers."""
    import copy

    from isaaclab_amd.env import ManagerBasedRLEnv
    from isaaclab_amd.events import EventManager
    from isaaclab_amd.robots import ANYMAL_C

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    push = {"func": "isaaclab.envs.mdp.events:push_by_setting_velocity", "mode": "interval", "interval_range_s": (0.03, 0.05), "is_global_time": True,
            "params": {"velocity_range": {"x": (-0.5, 0.5)}}}
    env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), events_cfg={"push_all": push, "startup_thing": {"func": "x:randomize_rigid_body_material", "mode": "startup", "params": {}}})
    assert env.event_manager.active_terms == {"interval": ["push_all"]} and env.event_manager.skipped_terms == ["startup_thing"]
    assert env.command_term is None and env.curriculum_manager is None
    env.reset()
    a = torch.zeros(64, 12, device="cuda:0")
    fired = 0
    for _ in range(6):
        before = env.sim_writes["root_vel"].clone()
        env.step(a)
        changed = (env.sim_writes["root_vel"] != before).any(dim=1)
        assert bool(changed.all()) or not bool(changed.any())  # one timer: every env at once, or none
        if bool(changed.all()):
            fired += 1
            d = env.sim_writes["root_vel"][:, 0] - env.feed["root_lin_vel_w"][:, 0]
            assert float(d.abs().max()) <= 0.5 + 1e-6 and torch.equal(env.sim_writes["root_vel"][:, 1], env.feed["root_lin_vel_w"][:, 1])
    assert 2 <= fired <= 4  # 0.02 s steps, a new interval of 0.03..0.05 s after every push
    with pytest.raises(RuntimeError, match="deferred step tail"):
        env.defer_step_tail = False
        env.step(a)
    env.close()
    # reset(env_ids=...) of an env with its own managers: _reset_idx for exactly those envs (commands resampled, counters back to 1,
    # reset events written), the others untouched; no interval event, no command compute
    from _util import OrchGolden

    og = OrchGolden()
    env = _orch_env(og, seed=5)
    env.reset()
    env.step(torch.zeros(og.N, 12, device="cuda:0"))
    ct = env.command_term
    ct.command_counter += 3
    before = {k: v.clone() for k, v in env.sim_writes.items()}
    cmd_before, tl_before = ct.vel_command_b.clone(), env.event_manager.get_term("push_robot").time_left.clone()
    ids = torch.tensor([2, 9, 40], device="cuda:0")
    _, extras = env.reset(env_ids=ids)
    others = torch.ones(og.N, dtype=torch.bool, device="cuda:0")
    others[ids] = False
    assert torch.equal(ct.command_counter[ids].cpu(), torch.ones(3, dtype=torch.long)) and bool((ct.command_counter[others] >= 4).all())
    assert torch.equal(ct.vel_command_b[others], cmd_before[others]) and not torch.equal(ct.vel_command_b[ids], cmd_before[ids])
    assert torch.equal(env.sim_writes["root_pose"][others], before["root_pose"][others]) and not torch.equal(env.sim_writes["root_pose"][ids], before["root_pose"][ids])
    assert torch.equal(env.event_manager.get_term("push_robot").time_left, tl_before)  # interval timers belong to step()
    assert "Metrics/base_velocity/error_vel_xy" in extras["log"] and "Curriculum/terrain_levels" in extras["log"]
    env.close()
    with pytest.raises(NotImplementedError, match="randomize_actuator_gains"):
        EventManager({"t": {"func": "isaaclab.envs.mdp.events:randomize_actuator_gains", "mode": "reset", "params": {}}}, 8, ANYMAL_C, "cuda:0")
    with pytest.raises(ValueError, match="interval_range_s"):
        EventManager({"t": dict(push, interval_range_s=None)}, 8, ANYMAL_C, "cuda:0")
    fx = copy.deepcopy(g.fixture)
    fx["env"]["curriculum"] = {"w": {"func": "isaaclab.envs.mdp.curriculums:modify_reward_weight", "params": {}}}
    with pytest.raises(NotImplementedError, match="modify_reward_weight"):
        ManagerBasedRLEnv(fx, state_feed=g.feed("cuda:0"), use_command_term=True, use_curriculum=True,
                          terrain_importer=__import__("isaaclab_amd.events", fromlist=["x"]).TerrainImporterState(
                              torch.zeros(2, 2, 3, device="cuda:0"), torch.zeros(64, dtype=torch.long, device="cuda:0"),
                              torch.zeros(64, dtype=torch.long, device="cuda:0"), 8.0))


@pytest.mark.parametrize("mode", ["fused-eager", "fused-graph", "generic-normalized"])
def test_runner_learn_modes(mode, tmp_path):
    """OnPolicyRunner.learn through the three rollout paths (train.py:167-183 surface), checkpoint round trip."""
    from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    feed = g.feed("cuda:0")  # 4 snapshots
    from isaaclab_amd.env import ManagerBasedRLEnv

    env = RslRlVecEnvWrapper(ManagerBasedRLEnv(g.fixture, state_feed=feed))
    cfg = dict(g.fixture["agent"], num_steps_per_env=8, empirical_normalization=(mode == "generic-normalized"))
    runner = OnPolicyRunner(env, cfg, log_dir=str(tmp_path), device="cuda:0", use_graph=(mode == "fused-graph"))
    assert runner._fusable() == (mode != "generic-normalized")
    p0 = runner.alg.bucket.flat.clone()
    runner.learn(2, init_at_random_ep_len=True)
    torch.cuda.synchronize()
    s = runner.alg.loss_dict()
    assert all(np.isfinite(v) for v in s.values()), s
    assert not torch.equal(p0, runner.alg.bucket.flat)  # parameters moved
    assert runner.alg.storage.observations.abs().sum() > 0 and runner.current_learning_iteration == 2
    assert int(runner.alg._adam[1]) == 2 * 5 * 4  # Adam steps = iterations x epochs x minibatches
    ck = str(tmp_path / "model.pt")
    runner.save(ck)
    before = runner.alg.bucket.flat.clone()
    m_before, v_before = runner.alg.bucket.exp_avg.clone(), runner.alg.bucket.exp_avg_sq.clone()
    assert float(m_before.abs().sum()) > 0
    runner.alg.bucket.flat.zero_()
    runner.alg.bucket.exp_avg.zero_()
    runner.alg.bucket.exp_avg_sq.zero_()
    runner.load(ck)
    assert torch.equal(before, runner.alg.bucket.flat)
    # Adam moments travel keyed by parameter name (independent of the bucket layout) and land where they came from
    assert torch.equal(m_before, runner.alg.bucket.exp_avg) and torch.equal(v_before, runner.alg.bucket.exp_avg_sq)
    policy = runner.get_inference_policy(device="cuda:0")
    obs_now = env.get_observations()[0]
    a = policy(obs_now)
    assert a.shape == (64, 12) and torch.isfinite(a).all()
    assert hasattr(runner.alg.policy, "actor") and hasattr(runner.alg.policy, "critic") and not runner.alg.policy.is_recurrent
    # the checkpoint has upstream's layout: a torch.optim.Adam can load its optimizer_state_dict as it is
    d = torch.load(ck, weights_only=True)
    opt = torch.optim.Adam(runner.alg.policy.parameters(), lr=1.0)
    opt.load_state_dict({k: v for k, v in d["optimizer_state_dict"].items() if k != "imx_adam_state"})
    assert abs(opt.param_groups[0]["lr"] - runner.alg.learning_rate) < 1e-12 and int(opt.state_dict()["state"][0]["step"]) == 40
    if mode == "generic-normalized":
        # the normaliser statistics travel with the checkpoint and the inference policy sees normalised observations
        assert {"obs_norm_state_dict", "privileged_obs_norm_state_dict"} <= set(d)
        mean0 = runner.obs_normalizer._mean.clone()
        assert float(mean0.abs().sum()) > 0
        expect = runner.alg.policy.act_inference(runner.obs_normalizer(obs_now))
        assert torch.equal(a, expect) and not torch.equal(a, runner.alg.policy.act_inference(obs_now))
        runner.obs_normalizer._mean.zero_()
        runner.load(ck)
        assert torch.equal(runner.obs_normalizer._mean, mean0)
        assert torch.equal(runner.get_inference_policy()(obs_now), a)
    # logging (upstream OnPolicyRunner.log; what scripts/benchmarks/benchmark_rsl_rl.py:220-231 reads back)
    import json as _json

    rows = [_json.loads(ln) for ln in open(tmp_path / "scalars.jsonl")]
    tags = {r["tag"] for r in rows}
    assert {"Perf/total_fps", "Perf/collection time", "Perf/learning_time", "Loss/value_function", "Loss/surrogate", "Loss/learning_rate",
            "Policy/mean_noise_std", "Episode_Reward/track_lin_vel_xy_exp", "Episode_Termination/time_out"} <= tags
    assert (tmp_path / "model_2.pt").exists()
    fps = [r["value"] for r in rows if r["tag"] == "Perf/total_fps"]
    assert len(fps) == 2 and all(v > 0 for v in fps)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_rollout_storage_is_self_consistent(use_graph):
    """After one collect() through the fused path (imx_mlp_infer, imx_policy_act, env kernels, imx_rollout_post; eager
    and as a hipGraph) every stored transition is consistent with the modules: mu = actor(obs), value = critic(obs),
    log-prob = Normal(mu, sigma).log_prob(action), obs[t+1] = what env.step returned, dones = terminated | time-outs."""
    from isaaclab_amd.env import ManagerBasedRLEnv
    from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    env = RslRlVecEnvWrapper(ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0")))
    cfg = dict(g.fixture["agent"], num_steps_per_env=8)
    runner = OnPolicyRunner(env, cfg, log_dir=None, device="cuda:0", use_graph=use_graph)
    assert runner._fusable()
    runner.train_mode()
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    env.episode_length_buf[::7] = int(env.max_episode_length) - 3  # some time-outs inside the rollout
    for _ in range(2):  # the second collect replays the captured graph
        runner.collect()
    torch.cuda.synchronize()
    st, pol = runner.alg.storage, runner.alg.policy
    T = runner.num_steps_per_env
    with torch.no_grad():
        for t in range(T):
            obs = st.observations[t]
            assert_close(st.mu[t], pol.actor(obs), 1e-5, f"mu[{t}]")
            assert_close(st.values[t], pol.critic(obs), 1e-5, f"values[{t}]")
            assert torch.equal(st.sigma[t], pol.std.expand_as(st.mu[t]))
            logp = torch.distributions.Normal(st.mu[t], st.sigma[t]).log_prob(st.actions[t]).sum(-1, keepdim=True)
            assert_close(st.actions_log_prob[t], logp, 1e-4, f"log-prob[{t}]")
            z = (st.actions[t] - st.mu[t]) / st.sigma[t]
            assert float(z.abs().max()) < 6.0 and 0.8 < float(z.std()) < 1.2  # a ~ N(mu, sigma)
        assert_close(runner.last_obs, env.unwrapped._obs, 0, "last obs")
    assert st.dones.sum() > 0 and set(st.dones.unique().tolist()) <= {0, 1}
    assert torch.isfinite(st.rewards).all() and float(st.rewards.abs().sum()) > 0
    # no privileged group: the critic's minibatch IS the policy's (no second buffer that the fused rollout would leave unfilled)
    assert st.privileged_observations is None
    batch = next(iter(st.mini_batch_generator(4, 1)))
    assert batch[1] is batch[0] and float(batch[0].abs().sum()) > 0


@pytest.mark.parametrize("case", ["flat-64-graph", "flat-64-eager", "rough-4096", "flat-2500-clip", "rough-4096-full-graph"])
def test_three_launch_rollout_step_equals_the_six_launch_split(case):
    """imx_mlp_infer_act (MLPs + PPO.act + ActionManager.process_action) and imx_terminations_rewards_rollout (+ the wrapper's dones,
    the time-out bootstrap, episode statistics; log sum in the step tail) against the split launches of round 2 (imx_mlp_infer,
    imx_policy_act, imx_action_process, imx_terminations_rewards, imx_observations, imx_rollout_post): two collects from identical
    states must leave BIT-IDENTICAL storages and env buffers.  64 envs take the 16-sample inference tiles, 2500 / 4096 the 32-sample
    ones (2500: a ragged last tile); one case clips the actions (RslRlVecEnvWrapper clip_actions)."""
    import copy

    from isaaclab_amd.env import ManagerBasedRLEnv
    from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper
    from isaaclab_amd.robots import ROBOTS
    from isaaclab_amd.state_feed import StateFeed

    def build(fuse):
        torch.manual_seed(3)
        if case.startswith("flat-64"):
            g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
            env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), noise_seed=11)
            agent, T = g.fixture["agent"], 8
        else:
            task = "Isaac-Velocity-Rough-Anymal-C-v0" if case.startswith("rough") else "Isaac-Velocity-Flat-Anymal-C-v0"
            g = Golden(task)
            N = 4096 if case.startswith("rough") else 2500
            mesh = g.mesh()
            ext = None
            if mesh is not None:
                ext = (float(np.abs(mesh[0][:, 0]).max()) - 2.0, float(np.abs(mesh[0][:, 1]).max()) - 2.0)
            feed = StateFeed(ROBOTS[g.fixture["robot"]], N, "cuda:0", seed=5, num_snapshots=4, extent_xy=ext)
            if "full" in case:  # bench.py --full-step: the env owns every producer around the physics step, all inside the rollout
                from bench import anydrive_like_net
                from isaaclab_amd.producers import ActuatorNetLSTM

                env = ManagerBasedRLEnv(g.fixture, state_feed=feed, terrain=mesh, noise_seed=11, own_managers=True, use_contact_sensor=True,
                                        use_articulation_update=True)
                lstm, head = anydrive_like_net("cuda:0")
                env.attach_actuator(ActuatorNetLSTM(N, 12, 80.0, 7.5, 120.0, lstm_layers=lstm, head=head, head_activation="softsign"))
            else:
                env = ManagerBasedRLEnv(g.fixture, state_feed=feed, terrain=mesh, noise_seed=11)
            agent, T = g.fixture["agent"], 4
        venv = RslRlVecEnvWrapper(env, clip_actions=0.8 if case.endswith("clip") else None)
        runner = OnPolicyRunner(venv, dict(agent, num_steps_per_env=T), log_dir=None, device="cuda:0", use_graph=case.endswith("graph"))
        runner.fuse_launches = fuse
        runner.train_mode()
        gen = torch.Generator().manual_seed(9)
        ep = torch.randint(0, int(venv.max_episode_length), (env.num_envs,), generator=gen)
        ep[::7] = int(venv.max_episode_length) - 2  # time-outs inside the rollout: the bootstrap term is exercised
        venv.episode_length_buf = ep.cuda()
        return env, runner

    out = {}
    for fuse in (False, True):
        env, runner = build(fuse)
        assert runner._fusable()
        for _ in range(2):  # with use_graph the second collect is a replay
            runner.collect()
        torch.cuda.synchronize()
        st = runner.alg.storage
        out[fuse] = {k: getattr(st, k).clone() for k in ("observations", "actions", "actions_log_prob", "mu", "sigma", "values", "rewards", "dones")}
        out[fuse].update(action=env._action.clone(), prev_action=env._prev_action.clone(), processed=env._processed_action.clone(),
                         cur_rew=runner._cur_reward_sum.clone(), cur_len=runner._cur_episode_length.clone(), last_obs=runner.last_obs.clone(),
                         ep_len=env.episode_length_buf.clone(), reset_ids=env.reset_env_ids.clone(), log_out=env._log_out.clone())
        out[fuse]["ep_stats"], out[fuse]["log_accum"] = runner._ep_stats.clone(), runner._log_accum.clone()
        if "full" in case:
            out[fuse].update(sim_root_vel=env.sim_writes["root_vel"].clone(), levels=env.terrain_importer.terrain_levels.clone(),
                             torque=env.actuator_net.applied_effort.clone(), command=env.command_term.vel_command_b.clone(),
                             air_time=env.contact_sensor.data.current_air_time.clone(), joint_acc=env.articulation.joint_acc.clone())
        env.close()
    a, b = out[False], out[True]
    for k in a:
        if k in ("ep_stats", "log_accum"):
            continue
        assert torch.equal(a[k], b[k]), f"{k}: fused and split rollouts differ"
    if "full" in case:
        assert float(a["sim_root_vel"].abs().sum()) > 0 and torch.isfinite(a["torque"]).all() and float(a["torque"].abs().sum()) > 0
    assert float(a["dones"].sum()) > 0 and float(a["log_accum"].abs().sum()) > 0
    assert torch.equal(a["log_accum"], b["log_accum"]), "per-iteration log sums"
    assert_close(b["ep_stats"], a["ep_stats"], 1e-5, "finished-episode statistics (float atomics: order differs)")
    if case.endswith("clip"):
        assert float(b["action"].abs().max()) <= float(np.float32(0.8))


@pytest.mark.gpu
def test_frame_table_from_the_step_kernel_equals_k_frame():
    """env.step() lets imx_terminations_rewards leave the per-env frame table (root-frame vectors, scanner yaw) for imx_observations
    (flag 4 skips k_frame): observations must be bit-identical to the ones computed with k_frame on the same state."""
    g = Golden("Isaac-Velocity-Rough-Anymal-C-v0")
    from isaaclab_amd.env import ManagerBasedRLEnv

    env = ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), terrain=g.mesh())
    env._noise_u = g.t("reset/noise_u").cuda()
    env.reset()
    for k in range(2):
        env._noise_u.copy_(g.t(f"step{k}/noise_u"))
        obs = env.step(g.t(f"step{k}/action").cuda())[0]["policy"].clone()  # frame written by the step kernel
        again = env._compute_observations().clone()                          # k_frame on the same state
        assert torch.equal(obs, again)


def test_kitchen_sink_matches_reference_golden():
    """Every remaining op of isaaclab.envs.mdp (SURVEY 8a "also present" rows: base_pos_z, root_*_w, joint_pos, joint_vel,
    joint_pos_limit_normalized; base_height_l2, body_lin_acc_l2, joint_vel_l2, joint_vel_limits, applied_torque_limits, action_l2,
    contact_forces, is_alive, is_terminated, is_terminated_term; bad_orientation, root_height_below_minimum, joint_vel_out_of_limit,
    joint_vel_out_of_manual_limit, joint_effort_out_of_limit, terrain_out_of_bounds, command_resample), a second observation group
    ("critic") scanning the same sensor, the height scanner's SensorBase gating (16 s old fp32 timestamps skip updates) and drift, and
    reward weights re-installed through get_term_cfg / set_term_cfg mid-run -- against the fixture the REAL reference produced."""
    from _util import KITCHEN

    g = Golden(KITCHEN)
    env = make_env(g)
    env.materialize_ray_hits = True
    N = g.N
    Dp, Dc = g.meta["obs_group_dims"]
    assert env.observation_manager.group_obs_dim == {"policy": (Dp,), "critic": (Dc,)}
    assert env.plan.scan_stateful
    env._noise_u = g.t("reset/noise_u").cuda()
    env._scan_drift_feed = g.t("reset/scan_drift").cuda()
    env.scanner_keep_all_hits = True  # the timestamps are overwritten below: the kernel cannot foresee which envs will skip step 0
    obs_dict, _ = env.reset()
    env.scanner_keep_all_hits = False
    assert_close(obs_dict["policy"], g.t("reset/obs"), FLOAT_TOL, "reset obs")
    assert_close(obs_dict["critic"], g.t("reset/obs/critic"), FLOAT_TOL, "reset critic obs")
    assert_close(env._ray_hits, g.t("reset/ray_hits_w"), FLOAT_TOL, "reset sensor hits")
    # sensors that have been running for 0 / 16.5 / 30 / 5 s (fp32 timestamps): inject like the fixture generator did
    env._scan_state[:, 0] = g.t("reset/scan_ts0").cuda()
    env._scan_state[:, 1] = g.t("reset/scan_ts0").cuda()
    env.episode_length_buf = g.t("reset/episode_length_buf")
    stale = 0
    for k in range(g.steps):
        tag = f"step{k}"
        for wc in g.meta["weight_changes"]:
            if wc[0] == k:  # envs/mdp/curriculums.py:32-36
                term_cfg = env.reward_manager.get_term_cfg(wc[1])
                term_cfg.weight = wc[2]
                env.reward_manager.set_term_cfg(wc[1], term_cfg)
        env._noise_u.copy_(g.t(f"{tag}/noise_u"))
        env._scan_drift_feed.copy_(g.t(f"{tag}/scan_drift"))
        obs_dict, rew, terminated, time_outs, extras = env.step(g.t(f"{tag}/action").cuda())
        torch.cuda.synchronize()
        assert torch.equal(terminated.cpu(), g.t(f"{tag}/terminated")) and torch.equal(time_outs.cpu(), g.t(f"{tag}/time_outs"))
        assert torch.equal(env.reset_env_ids.cpu(), g.t(f"{tag}/reset_env_ids"))
        for name in g.meta["termination_terms"]:
            assert torch.equal(env.termination_manager.get_term(name).cpu(), g.t(f"{tag}/term_dones/{name}")), name
        assert torch.equal(env.episode_length_buf.cpu(), g.t(f"{tag}/episode_length_buf"))
        assert_close(rew, g.t(f"{tag}/reward"), FLOAT_TOL, "reward")
        assert_close(env.reward_manager._step_reward, g.t(f"{tag}/step_reward"), FLOAT_TOL, "step_reward")
        for name in g.meta["reward_terms"]:
            assert_close(env.reward_manager._episode_sums[name], g.t(f"{tag}/episode_sums/{name}"), FLOAT_TOL, name)
        # the sensor: which envs refreshed (timestamps bit-exact), what they hold
        cur = env._scan_state.cpu()
        if k == 1:  # a second compute() within the step (user code) repeats the step's decision instead of advancing the sensor clock
            before = {k_: v.clone() for k_, v in obs_dict.items()}
            again = env.observation_manager.compute()
            assert all(torch.equal(again[k_], before[k_]) for k_ in before) and torch.equal(env._scan_state.cpu()[:, :7], cur[:, :7])
        assert torch.equal(cur[:, 0], g.t(f"{tag}/scan_timestamp")), "sensor timestamps"
        assert torch.equal(cur[:, 1], g.t(f"{tag}/scan_timestamp_last_update")), "sensor last-update stamps"
        stale += int((cur[:, 0] != cur[:, 1]).sum())
        assert_close(cur[:, 5], g.t(f"{tag}/sensor_pos_w")[:, 2], 1e-6, "sensor data.pos_w z")
        assert_close(env._ray_hits[..., 2], g.t(f"{tag}/ray_hits_w")[..., 2], FLOAT_TOL, "sensor hit heights (stale rows keep the old cast)")
        assert_close(obs_dict["policy"], g.t(f"{tag}/obs"), FLOAT_TOL, "obs")
        assert_close(obs_dict["critic"], g.t(f"{tag}/obs/critic"), FLOAT_TOL, "critic obs")
        for key, v in g.log(k).items():
            got = float(extras["log"][key])
            assert abs(got - v) <= FLOAT_TOL * max(1.0, abs(v)), (key, got, v)
    assert stale > 10
    assert env.reward_manager.get_term_cfg("action_l2").weight == -0.02
    with pytest.raises(ValueError):
        env.reward_manager.set_term_cfg("no_such_term", None)
    env.close()


def test_env_reset_after_a_step_resets_the_stateful_scanner():
    """reset(env_ids) after at least one step goes through k_frame with the step stamp k_term_rew left (a "repeated" call of that step):
    the sensor of those envs must still be reset -- timers to zero, outdated, a new drift (manager_based_env.py:264-315 -> scene.reset ->
    SensorBase.reset :182-194 + RayCaster.reset :107-114) -- and the others left alone."""
    from _util import KITCHEN

    g = Golden(KITCHEN)
    env = make_env(g)
    N = g.N
    env._noise_u = g.t("reset/noise_u").cuda()
    env._scan_drift_feed = g.t("reset/scan_drift").cuda()
    env.reset()
    env._scan_state[:, 0] = g.t("reset/scan_ts0").cuda()
    env._scan_state[:, 1] = g.t("reset/scan_ts0").cuda()
    env.episode_length_buf = g.t("reset/episode_length_buf")
    env._noise_u.copy_(g.t("step0/noise_u"))
    env._scan_drift_feed.copy_(g.t("step0/scan_drift"))
    env.step(g.t("step0/action").cuda())
    before = env._scan_state.cpu().clone()
    ids = torch.tensor([1, 7, 30, N - 1])
    ids = ids[before[ids, 0] > 0.0]  # envs whose sensor clock is running (not reset by the step itself)
    assert len(ids) >= 2
    new_drift = torch.full((N, 3), 0.0)
    new_drift[ids] = torch.tensor([0.011, -0.007, 0.003])
    env._scan_drift_feed.copy_(new_drift)
    obs, _ = env.reset(env_ids=ids.cuda())
    after = env._scan_state.cpu()
    others = torch.ones(N, dtype=torch.bool)
    others[ids] = False
    assert torch.equal(after[ids, 0], torch.zeros(len(ids))) and torch.equal(after[ids, 1], torch.zeros(len(ids))), "sensor timers of the reset envs"
    assert torch.equal(after[ids, 2:5], new_drift[ids]), "a new drift for the reset envs"
    assert torch.equal(after[others][:, :5], before[others][:, :5]), "the other sensors keep their clock and drift"
    # the reset envs were cast from the drifted pose: data.pos_w z = root z + drift z
    root_z = env.feed["root_pos_w"][ids.cuda(), 2].cpu()
    assert_close(after[ids, 5], root_z + new_drift[ids, 2], 1e-6, "sensor data.pos_w z after the reset")
    # and a full env.reset() after steps resets every sensor
    env._scan_drift_feed.copy_(torch.full((N, 3), 0.002))
    env.reset()
    after = env._scan_state.cpu()
    assert torch.equal(after[:, 0], torch.zeros(N)) and torch.equal(after[:, 2:5], torch.full((N, 3), 0.002))
    env.close()


# ---- Python-evaluated ("EXTERNAL") terms: user functions the term compiler does not know (ManagerBase term contract,
# managers/manager_base.py:278-395: func(env, **params) -> Tensor[N, ...], SceneEntityCfg parameters resolved)
def _user_reward(env, asset_cfg, gain: float):
    return gain * torch.sum(torch.abs(env.scene["robot"].data.joint_vel[:, asset_cfg.joint_ids]), dim=1)


def _user_termination(env, limit: float):
    return env.scene["robot"].data.root_lin_vel_b[:, 0] > limit


def _user_obs(env, asset_cfg):
    d = env.scene["robot"].data
    return torch.cat([d.joint_pos[:, asset_cfg.joint_ids] ** 2, d.projected_gravity_b[:, 2:3]], dim=1)


def _user_modifier(x, gain: float):
    return x * gain + 1.0


def test_python_evaluated_terms_step_end_to_end():
    """One reward, one termination and one observation term the compiler does not know are routed to IMX_*_EXTERNAL, evaluated by
    calling the Python function each step, and folded into the fused kernels (weighting, reset bookkeeping, noise / clip / scale).
    Checked against the CPU oracle with the same three functions written as oracle terms."""
    import copy

    from isaaclab_amd.env import ManagerBasedRLEnv
    from oracle.mdp_oracle import OracleEnv

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    fx = copy.deepcopy(g.fixture)
    ent = {"name": "robot", "joint_names": [".*KFE"], "joint_ids": "slice(None, None, None)", "body_names": None, "body_ids": "slice(None, None, None)",
           "preserve_order": False}
    fx["env"]["rewards"]["user_rew"] = {"func": _user_reward, "params": {"asset_cfg": ent, "gain": 0.5}, "weight": -0.25}
    fx["env"]["terminations"]["user_term"] = {"func": _user_termination, "params": {"limit": 0.9}, "time_out": False}
    fx["env"]["observations"]["policy"]["user_obs"] = {
        "func": _user_obs, "params": {"asset_cfg": ent}, "_dim": 5, "noise": {"func": "isaaclab.utils.noise.noise_model:uniform_noise",
        "n_min": -0.1, "n_max": 0.1, "operation": "add"}, "clip": [-0.5, 2.0], "scale": 2.0,
        "modifiers": [{"func": _user_modifier, "params": {"gain": 0.5}}]}
    env = ManagerBasedRLEnv(fx, state_feed=g.feed("cuda:0"))
    assert (env.plan.n_ext_rew, env.plan.n_ext_term, env.plan.n_ext_obs) == (1, 1, 5)
    D = env.plan.obs_dim
    assert D == g.meta["obs_dim"] + 5

    class Orc(OracleEnv):  # the same three user terms on the CPU side (the oracle keys terms on "module:function" strings)
        kfe = [i for i, n in enumerate(g.robot.joint_names) if n.endswith("KFE")]

        def _reward(self, fn, p):
            if fn.endswith(":_user_reward"):
                return p["gain"] * torch.sum(torch.abs(self.state("joint_vel")[:, self.kfe]), dim=1)
            return super()._reward(fn, p)

        def _termination(self, fn, p):
            if fn.endswith(":_user_termination"):
                return self.root_lin_vel_b[:, 0] > p["limit"]
            return super()._termination(fn, p)

        def _obs_term(self, fn, p):
            if fn.endswith(":_user_obs"):
                return torch.cat([self.state("joint_pos")[:, self.kfe] ** 2, self.projected_gravity_b[:, 2:3]], dim=1)
            return super()._obs_term(fn, p)

        def _modifier(self, term, idx, m, obs):
            if m["func"].endswith(":_user_modifier"):
                return _user_modifier(obs, **m["params"])
            return super()._modifier(term, idx, m, obs)

    cpu_feed = g.feed("cpu")
    fo = copy.deepcopy(fx["env"])
    fo["rewards"]["user_rew"]["func"] = "user:_user_reward"
    fo["terminations"]["user_term"]["func"] = "user:_user_termination"
    fo["observations"]["policy"]["user_obs"]["func"] = "user:_user_obs"
    fo["observations"]["policy"]["user_obs"]["modifiers"][0]["func"] = "user:_user_modifier"
    orc = Orc(fo, g.robot.joint_names, g.robot.body_names, g.N, cpu_feed.__getitem__, g.meta["gravity_dir"])
    gen = torch.Generator().manual_seed(5)
    u = torch.rand(g.N, D, generator=gen)
    env._noise_u = u.cuda()
    obs, _ = env.reset()
    assert_close(obs["policy"], orc.compute_observations(u), FLOAT_TOL, "reset obs with a Python-evaluated term")
    ep = g.t("reset/episode_length_buf")
    env.episode_length_buf = ep
    orc.episode_length_buf[:] = ep
    fired = 0
    for k in range(g.steps):
        a = g.t(f"step{k}/action")
        u = torch.rand(g.N, D, generator=gen)
        env._noise_u.copy_(u)
        obs, rew, term, tout, _ = env.step(a.cuda())
        orc.process_action(a)
        cpu_feed.advance()
        out = orc.post_physics_step(u)
        assert torch.equal(term.cpu(), out["terminated"]) and torch.equal(env.reset_env_ids.cpu(), out["reset_env_ids"])
        assert torch.equal(env.termination_manager.get_term("user_term").cpu(), orc.term_dones["user_term"])
        fired += int(orc.term_dones["user_term"].sum())
        assert_close(rew, out["reward"], FLOAT_TOL, "reward")
        assert_close(env.reward_manager._episode_sums["user_rew"], orc.episode_sums["user_rew"], FLOAT_TOL, "episode sum of the user reward")
        assert_close(obs["policy"], out["obs"], FLOAT_TOL, "obs")
    assert fired > 0 and float(orc.episode_sums["user_rew"].abs().sum()) > 0
    env.close()


class _StreakReward:
    """A STATEFUL class reward term (managers/manager_base.py:28-115 contract): how many steps in a row the knee joints moved fast."""

    instances = 0

    def __init__(self, cfg, env):
        type(self).instances += 1
        self.cfg, self._env = cfg, env
        self.streak = torch.zeros(env.num_envs, device=env.device)
        self.reset_calls = []

    def reset(self, env_ids=None):
        self.reset_calls.append(None if env_ids is None else torch.as_tensor(env_ids).cpu().clone())
        self.streak[slice(None) if env_ids is None else env_ids] = 0.0

    def __call__(self, env, asset_cfg, threshold: float):
        fast = torch.sum(torch.abs(env.scene["robot"].data.joint_vel[:, asset_cfg.joint_ids]), dim=1) > threshold
        self.streak = torch.where(fast, self.streak + 1.0, torch.zeros_like(self.streak))
        return self.streak.clone()


class _EmaObs:
    """A stateful class observation term: moving average of the projected gravity, restarted at reset."""

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self.ema = torch.zeros(env.num_envs, 3, device=env.device)

    def reset(self, env_ids=None):
        self.ema[slice(None) if env_ids is None else env_ids] = 0.0

    def __call__(self, env, alpha: float):
        self.ema = alpha * env.scene["robot"].data.projected_gravity_b + (1.0 - alpha) * self.ema
        return self.ema.clone()


def test_class_based_python_evaluated_terms():
    """Class terms on the Python-evaluated route: built once with (cfg, env), called every step, ``reset(env_ids)`` with the ids of the
    step's resets between the reward and the observation pass (manager_base.py:324-327,393-395; reward_manager.py:123-124;
    observation_manager.py:224) -- against the CPU oracle carrying the same two state machines."""
    import copy

    from isaaclab_amd.env import ManagerBasedRLEnv
    from oracle.mdp_oracle import OracleEnv

    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    fx = copy.deepcopy(g.fixture)
    ent = {"name": "robot", "joint_names": [".*KFE"], "joint_ids": "slice(None, None, None)", "body_names": None, "body_ids": "slice(None, None, None)",
           "preserve_order": False}
    fx["env"]["rewards"]["streak"] = {"func": _StreakReward, "params": {"asset_cfg": ent, "threshold": 3.0}, "weight": 0.5}
    fx["env"]["observations"]["policy"]["ema"] = {"func": _EmaObs, "params": {"alpha": 0.25}, "_dim": 3, "scale": 2.0}
    _StreakReward.instances = 0
    env = ManagerBasedRLEnv(fx, state_feed=g.feed("cuda:0"))
    assert _StreakReward.instances == 1 and len(env._class_terms) == 2
    rew_inst = env._class_terms[0]
    assert isinstance(rew_inst, _StreakReward) and rew_inst.cfg.weight == 0.5 and rew_inst.cfg.params["threshold"] == 3.0
    D = env.plan.obs_dim
    assert D == g.meta["obs_dim"] + 3

    class Orc(OracleEnv):
        kfe = [i for i, n in enumerate(g.robot.joint_names) if n.endswith("KFE")]

        def _reward(self, fn, p):
            if fn.endswith(":_StreakReward"):
                fast = torch.sum(torch.abs(self.state("joint_vel")[:, self.kfe]), dim=1) > p["threshold"]
                self.streak = torch.where(fast, self.streak + 1.0, torch.zeros_like(self.streak))
                return self.streak.clone()
            return super()._reward(fn, p)

        def _obs_term(self, fn, p):
            if fn.endswith(":_EmaObs"):
                self.ema = p["alpha"] * self.projected_gravity_b + (1.0 - p["alpha"]) * self.ema
                return self.ema.clone()
            return super()._obs_term(fn, p)

    cpu_feed = g.feed("cpu")
    fo = copy.deepcopy(fx["env"])
    fo["rewards"]["streak"]["func"] = "user:_StreakReward"
    fo["observations"]["policy"]["ema"]["func"] = "user:_EmaObs"
    orc = Orc(fo, g.robot.joint_names, g.robot.body_names, g.N, cpu_feed.__getitem__, g.meta["gravity_dir"])
    orc.streak, orc.ema = torch.zeros(g.N), torch.zeros(g.N, 3)

    def orc_reset_class_terms(ids):  # _reset_idx -> RewardManager.reset / ObservationManager.reset -> term.reset(env_ids)
        if len(ids) > 0:
            orc.streak[ids] = 0.0
            orc.ema[ids] = 0.0

    orc.pre_obs_hook = orc_reset_class_terms
    gen = torch.Generator().manual_seed(5)
    u = torch.rand(g.N, D, generator=gen)
    env._noise_u = u.cuda()
    obs, _ = env.reset()
    assert rew_inst.reset_calls == [None]  # env.reset(): every env
    assert_close(obs["policy"], orc.compute_observations(u), FLOAT_TOL, "reset obs with a class term")
    ep = g.t("reset/episode_length_buf")
    env.episode_length_buf = ep
    orc.episode_length_buf[:] = ep
    n_reset_calls = 0
    for k in range(g.steps):
        a = g.t(f"step{k}/action")
        u = torch.rand(g.N, D, generator=gen)
        env._noise_u.copy_(u)
        obs, rew, term, tout, _ = env.step(a.cuda())
        orc.process_action(a)
        cpu_feed.advance()
        out = orc.post_physics_step(u)
        assert torch.equal(env.reset_env_ids.cpu(), out["reset_env_ids"])
        if len(out["reset_env_ids"]) > 0:
            n_reset_calls += 1
            assert torch.equal(rew_inst.reset_calls[-1], out["reset_env_ids"])  # the ids of THIS step's resets, ascending
        assert len(rew_inst.reset_calls) == 1 + n_reset_calls
        assert_close(rew, out["reward"], FLOAT_TOL, "reward")
        assert_close(rew_inst.streak, orc.streak, 0.0, "class reward state")
        assert_close(env.reward_manager._episode_sums["streak"], orc.episode_sums["streak"], FLOAT_TOL, "episode sum of the class reward")
        assert_close(obs["policy"], out["obs"], FLOAT_TOL, "obs")
    assert n_reset_calls > 0 and float(orc.streak.sum()) > 0
    env.close()



@pytest.mark.parametrize("mode", ["fused-eager", "fused-graph", "generic"])
def test_runner_with_a_critic_observation_group(mode):
    """An env with a privileged ("critic") observation group (isaaclab_rl/rsl_rl/vecenv_wrapper.py:71-79): the critic network and the
    storage's privileged observations take THAT group in every rollout path -- never the policy observations."""
    from _util import KITCHEN
    from isaaclab_amd.env import ManagerBasedRLEnv
    from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

    g = Golden(KITCHEN)
    env = RslRlVecEnvWrapper(ManagerBasedRLEnv(g.fixture, state_feed=g.feed("cuda:0"), terrain=g.mesh()))
    Dp, Dc = g.meta["obs_group_dims"]
    assert (env.num_obs, env.num_privileged_obs) == (Dp, Dc)
    cfg = dict(g.fixture["agent"], num_steps_per_env=6)  # 6 snapshots in the recorded feed
    runner = OnPolicyRunner(env, cfg, device="cuda:0", use_graph=(mode == "fused-graph"))
    if mode == "generic":
        runner._fusable = lambda: False
    assert runner.privileged_obs_type == "critic"
    assert runner.alg.policy.critic[0].in_features == Dc and runner.alg.policy.actor[0].in_features == Dp
    st = runner.alg.storage
    assert st.privileged_observations.shape == (6, 64, Dc)
    runner.collect()
    torch.cuda.synchronize()
    with torch.no_grad():
        v = runner.alg.policy.critic(st.privileged_observations.flatten(0, 1)).view(6, 64, 1)
    assert_close(st.values, v, 1e-4, "stored values = critic(privileged observations)")
    crit_now = env.unwrapped.obs_buf["critic"]
    assert torch.equal(runner.last_critic_obs, crit_now) and crit_now.shape[1] == Dc
    with torch.no_grad():  # and the policy side: mu = actor(stored policy observations)
        mu = runner.alg.policy.actor(st.observations.flatten(0, 1)).view(6, 64, -1)
    assert_close(st.mu, mu, 1e-4, "stored action means = actor(policy observations)")
    # the stored critic rows are critic-group rows: their velocity-command columns equal the policy group's (no noise on either)
    names = g.meta["obs_terms"]
    assert float(st.privileged_observations.abs().sum()) > 0
    st.clear()
    runner.learn(2)
    torch.cuda.synchronize()
    assert all(np.isfinite(x) for x in runner.alg.loss_dict().values())


def test_seed_reseeds_the_in_kernel_generators_and_spaces_exist():
    """isaaclab_tasks/test/test_environment_determinism.py:57-66 ("same seed, same rollout") through ``env.seed()`` / ``reset(seed=)``
    given AFTER construction: the in-kernel observation-noise stream follows the seed.  Plus the gym spaces of
    manager_based_rl_env.py:319-345."""
    g = Golden("Isaac-Velocity-Flat-Anymal-C-v0")
    a = g.t("step0/action").cuda()

    def rollout(seed):
        env = make_env(g)
        env.reset(seed=seed)
        out = [env.step(a)[0]["policy"].clone() for _ in range(2)]
        env.close()
        return out

    r1, r2, r3 = rollout(7), rollout(7), rollout(8)
    assert all(torch.equal(x, y) for x, y in zip(r1, r2))
    assert not torch.equal(r1[0], r3[0])  # another seed, another noise stream
    env = make_env(g)
    assert tuple(env.single_action_space.shape) == (12,) and tuple(env.action_space.shape) == (64, 12)
    assert tuple(env.single_observation_space["policy"].shape) == (48,) and tuple(env.observation_space["policy"].shape) == (64, 48)
    env.close()
