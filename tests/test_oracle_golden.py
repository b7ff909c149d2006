"""CPU: the oracle (oracle/mdp_oracle.py) against fixtures produced by the REAL reference managers/terms
(oracle/gen_golden.py).  This is what pins the oracle; masks/ids bit-exact, floats <= 1e-6."""

import numpy as np
import pytest
import torch

import os

from _util import GOLDEN, TASKS, Golden, assert_close
from oracle.mdp_oracle import OracleEnv


def make_oracle(g: Golden, feed):
    return OracleEnv(g.fixture["env"], g.robot.joint_names, g.robot.body_names, g.N, feed.__getitem__,
                     gravity_dir=g.meta["gravity_dir"])


@pytest.mark.parametrize("task", TASKS)
def test_oracle_matches_reference(task):
    g = Golden(task)
    feed = g.feed()
    env = make_oracle(g, feed)
    has_scan = g.mesh() is not None
    if has_scan:
        env.ray_hits_w = g.t("reset/ray_hits_w")
    env.reset_action_terms()  # ManagerBasedEnv.reset -> _reset_idx(all) -> ActionManager.reset
    obs = env.compute_observations(g.t("reset/noise_u"))
    assert_close(obs, g.t("reset/obs"), 1e-6, "reset obs")
    env.episode_length_buf[:] = g.t("reset/episode_length_buf")
    for k in range(g.steps):
        tag = f"step{k}"
        env.process_action(g.t(f"{tag}/action"))
        assert_close(env.processed_actions, g.t(f"{tag}/processed_actions"), 1e-6, "processed_actions")
        feed.advance()
        if has_scan:
            env.ray_hits_w = g.t(f"{tag}/ray_hits_w")
        out = env.post_physics_step(g.t(f"{tag}/noise_u"))
        assert torch.equal(out["reset_buf"], g.t(f"{tag}/reset_buf"))
        assert torch.equal(out["terminated"], g.t(f"{tag}/terminated"))
        assert torch.equal(out["time_outs"], g.t(f"{tag}/time_outs"))
        assert torch.equal(out["reset_env_ids"], g.t(f"{tag}/reset_env_ids"))
        for name in g.meta["termination_terms"]:
            assert torch.equal(env.term_dones[name], g.t(f"{tag}/term_dones/{name}"))
        assert_close(out["reward"], g.t(f"{tag}/reward"), 1e-6, "reward")
        assert_close(out["step_reward"], g.t(f"{tag}/step_reward"), 1e-6, "step_reward")
        for name in g.meta["reward_terms"]:
            assert_close(env.episode_sums[name], g.t(f"{tag}/episode_sums/{name}"), 1e-6, f"episode_sums/{name}")
        assert torch.equal(env.episode_length_buf, g.t(f"{tag}/episode_length_buf"))
        assert_close(env.action, g.t(f"{tag}/action_after_reset"), 0, "action after reset")
        assert_close(out["obs"], g.t(f"{tag}/obs"), 1e-6, "obs")
        ref_log = g.log(k)
        for key, v in ref_log.items():
            assert abs(out["log"][key] - v) <= 1e-6 * max(1.0, abs(v)), key


def test_math_helpers_match_reference():
    import os

    from _util import GOLDEN
    import oracle.mdp_oracle as m

    z = np.load(os.path.join(GOLDEN, "math.npz"))
    q, v, ang = (torch.from_numpy(z[k]) for k in ("q", "v", "ang"))
    for name, got in {
        "quat_rotate_inverse": m.quat_rotate_inverse(q, v), "quat_rotate": m.quat_rotate(q, v),
        "quat_apply": m.quat_apply(q, v), "yaw_quat": m.yaw_quat(q), "quat_apply_yaw": m.quat_apply_yaw(q, v),
        "wrap_to_pi": m.wrap_to_pi(ang), "normalize": m.normalize(v),
        "convert_quat_to_wxyz": m.convert_quat(q, "wxyz"), "convert_quat_to_xyzw": m.convert_quat(q, "xyzw"),
        "scale_transform": m.scale_transform(v, torch.from_numpy(z["lower"]), torch.from_numpy(z["upper"])),
    }.items():
        assert torch.equal(got, torch.from_numpy(z[name])), name  # same ops, same machine: bit-exact


def test_events_oracle_matches_reference():
    """oracle/events_oracle.py against tests/golden/events.npz (real reference event terms + terrain curriculum)."""
    import json

    from oracle import events_oracle as eo

    z = np.load(os.path.join(GOLDEN, "events.npz"))
    meta = json.loads(str(z["meta"]))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    mask = t("mask")
    pose, vel = eo.reset_root_state_uniform(t("default_root_state"), t("env_origins"), meta["pose_range"], meta["velocity_range"],
                                            t("root/u_pose"), t("root/u_vel"))
    assert_close(pose[mask], t("root/pose_out")[mask], 1e-6, "root pose")
    assert_close(vel[mask], t("root/vel_out")[mask], 1e-6, "root velocity")
    for tag in ("scale", "scale2", "offset"):
        r = z[f"joints_{tag}/ranges"]
        p, v = eo.reset_joints(t("default_joint_pos"), t("default_joint_vel"), t("soft_joint_pos_limits"), t("soft_joint_vel_limits"),
                               (float(r[0]), float(r[1])), (float(r[2]), float(r[3])), t(f"joints_{tag}/u_pos"), t(f"joints_{tag}/u_vel"),
                               tag == "offset")
        assert_close(p[mask], t(f"joints_{tag}/pos_out")[mask], 1e-6, f"joint pos {tag}")
        assert_close(v[mask], t(f"joints_{tag}/vel_out")[mask], 1e-6, f"joint vel {tag}")
    pv = eo.push_by_setting_velocity(t("root_vel_w"), meta["push_range"], t("push/u"))
    assert_close(pv[mask], t("push/vel_out")[mask], 1e-6, "push")
    r4 = z["ext/ranges"]
    fo, to = eo.apply_external_force_torque((float(r4[0]), float(r4[1])), (float(r4[2]), float(r4[3])), t("ext/u_force"), t("ext/u_torque"))
    assert_close(fo[mask], t("ext/forces")[mask], 1e-6, "external forces")
    assert_close(to[mask], t("ext/torques")[mask], 1e-6, "external torques")
    lv, og, mean = eo.terrain_levels_vel(mask, t("curr/root_pos_w"), t("curr/env_origins_in"), t("curr/command"), t("curr/terrain_origins"),
                                         t("curr/levels_in"), t("curr/types"), meta["terrain_size"], meta["max_episode_length_s"],
                                         t("curr/randint"))
    assert torch.equal(lv, t("curr/levels_out"))
    assert torch.equal(og, t("curr/env_origins_out"))
    assert abs(float(mean) - float(z["curr/mean_level"])) < 1e-6


@pytest.mark.parametrize("task", [t for t in TASKS if "Rough" in t])
def test_rough_fixture_mesh_is_the_generator_mesh_at_head(task):
    """A fixture with a terrain stores the sha256 of the mesh it was generated on (oracle/gen_golden.py::mesh_sha256).  The mesh in
    the .npz must carry that hash, and so must the terrain generator at HEAD with the recorded arguments: a change to
    ``make_rough_terrain`` that is not followed by regenerating the fixtures fails here instead of stranding them."""
    import hashlib

    from isaaclab_amd.terrain import make_rough_terrain

    def sha(v, t):
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(v, np.float32).tobytes())
        h.update(np.ascontiguousarray(t, np.uint32).tobytes())
        return h.hexdigest()

    g = Golden(task)
    v, t = g.mesh()
    assert sha(v, t) == g.meta["mesh_sha256"]
    v2, t2, _ = make_rough_terrain(**g.meta["terrain_args"])
    assert sha(v2, t2) == g.meta["mesh_sha256"], "terrain generator changed: re-run oracle/gen_golden.py for the rough tasks"
    # the small terrain must contain box-primitive (general) cells as well as height-field (lattice) cells
    assert len(t) < 2 * 81 * 81 * 6, "expected a mixed mesh, not six all-height-field tiles"


def test_oracle_matches_reference_group_shapes():
    """ObservationManager's other group shapes (observation_manager.py:320-335), fixture from the real manager: a group handed out as
    a dict of terms (concatenate_terms=False) with un-flattened (N, H, d) history terms next to flattened and plain ones, and a
    concatenated group whose terms keep their history axis, (N, H, sum d)."""
    from _util import SHAPES, assert_groups_close

    g = Golden(SHAPES)
    assert g.meta["obs_group_concatenate"] == {"policy": True, "terms": False, "stack": True}
    feed = g.feed()
    env = make_oracle(g, feed)
    assert_groups_close(env.compute_observation_groups(g.t("reset/noise_u")), g, "reset", 1e-6)
    env.episode_length_buf[:] = g.t("reset/episode_length_buf")
    for k in range(g.steps):
        tag = f"step{k}"
        env.process_action(g.t(f"{tag}/action"))
        feed.advance()
        out = env.post_physics_step(g.t(f"{tag}/noise_u"))
        assert torch.equal(out["reset_env_ids"], g.t(f"{tag}/reset_env_ids"))
        assert_groups_close(out["obs_groups"], g, tag, 1e-6)
    assert out["obs_groups"]["stack"].shape == (g.N, 2, 9) and out["obs_groups"]["terms"]["joint_vel"].shape == (g.N, 2, 12)


def test_oracle_matches_reference_kitchen_sink():
    """The kitchen-sink fixture (oracle/gen_golden.py::kitchen_cfg, generated by the REAL managers, terms, RayCaster/SensorBase and
    the real ``modify_reward_weight``): every op of isaaclab.envs.mdp the four task configs do not use, two observation groups, the
    height scanner's update-period gating (fp32 timestamps of 16 s old sensors skip updates) and drift, reward weights changed
    mid-run."""
    import copy

    from _util import KITCHEN, set_reward_weight
    from oracle.mdp_oracle import OracleScanner

    g = Golden(KITCHEN)
    assert g.meta["real_scanner"] and g.meta["obs_groups"] == ["policy", "critic"]
    feed = g.feed()
    cfg_env = copy.deepcopy(g.fixture["env"])
    env = OracleEnv(cfg_env, g.robot.joint_names, g.robot.body_names, g.N, feed.__getitem__, gravity_dir=g.meta["gravity_dir"])
    sc_cfg = cfg_env["scene"]["height_scanner"]
    R = g.t("reset/ray_hits_fresh").shape[1]
    scan = OracleScanner(g.N, R, sc_cfg["update_period"], tuple(sc_cfg["drift_range"]))

    def refresh(tag):
        scan.refresh(feed["root_pos_w"], g.t(f"{tag}/ray_hits_fresh"))
        env.ray_hits_w, env.sensor_pos_w = scan.ray_hits_w, scan.pos_w
        assert torch.equal(scan.pos_w, g.t(f"{tag}/sensor_pos_w")), tag
        # the same rows were refreshed (1e-6: torch's vectorised sin/cos of yaw_quat differ in the last bit between the all-env batch the
        # "fresh" hits were cast with and the outdated-env subset the real sensor cast)
        assert_close(scan.ray_hits_w, g.t(f"{tag}/ray_hits_w"), 1e-6, f"{tag} sensor hits")
        assert torch.equal(scan.timestamp, g.t(f"{tag}/scan_timestamp")) and torch.equal(scan.timestamp_last_update, g.t(f"{tag}/scan_timestamp_last_update"))

    scan.reset(None, drift=g.t("reset/scan_drift"))
    refresh("reset")
    obs = env.compute_observation_groups(g.t("reset/noise_u"))
    assert_close(obs["policy"], g.t("reset/obs"), 1e-6, "reset obs")
    assert_close(obs["critic"], g.t("reset/obs/critic"), 1e-6, "reset critic obs")
    scan.timestamp[:] = g.t("reset/scan_ts0")
    scan.timestamp_last_update[:] = g.t("reset/scan_ts0")
    env.episode_length_buf[:] = g.t("reset/episode_length_buf")
    stale_seen = 0
    for k in range(g.steps):
        tag = f"step{k}"
        for wc in g.meta["weight_changes"]:
            if wc[0] == k:
                set_reward_weight(cfg_env, wc[1], wc[2])
        env.process_action(g.t(f"{tag}/action"))
        feed.advance()
        for _ in range(cfg_env["decimation"]):  # scene.update(physics_dt) in the decimation loop
            scan.update(cfg_env["sim"]["dt"])

        def after_reset(reset_env_ids, tag=tag):
            if len(reset_env_ids) > 0:
                scan.reset(reset_env_ids, drift=g.t(f"{tag}/scan_drift"))
            refresh(tag)

        env.pre_obs_hook = after_reset
        out = env.post_physics_step(g.t(f"{tag}/noise_u"))
        stale_seen += int((scan.timestamp != scan.timestamp_last_update).sum())
        assert torch.equal(out["reset_buf"], g.t(f"{tag}/reset_buf"))
        assert torch.equal(out["terminated"], g.t(f"{tag}/terminated")) and torch.equal(out["time_outs"], g.t(f"{tag}/time_outs"))
        assert torch.equal(out["reset_env_ids"], g.t(f"{tag}/reset_env_ids"))
        for name in g.meta["termination_terms"]:
            assert torch.equal(env.term_dones[name], g.t(f"{tag}/term_dones/{name}")), name
        assert_close(out["reward"], g.t(f"{tag}/reward"), 1e-6, "reward")
        assert_close(out["step_reward"], g.t(f"{tag}/step_reward"), 1e-6, "step_reward")
        for name in g.meta["reward_terms"]:
            assert_close(env.episode_sums[name], g.t(f"{tag}/episode_sums/{name}"), 1e-6, f"episode_sums/{name}")
        assert_close(out["obs_groups"]["policy"], g.t(f"{tag}/obs"), 1e-6, "obs")
        assert_close(out["obs_groups"]["critic"], g.t(f"{tag}/obs/critic"), 1e-6, "critic obs")
        for key, v in g.log(k).items():
            assert abs(out["log"][key] - v) <= 1e-6 * max(1.0, abs(v)), key
    assert stale_seen > 10  # the fixture really exercises skipped sensor updates


def test_orchestration_oracle_matches_the_real_reset_idx_and_managers():
    """oracle/orchestration_oracle.py against 48 steps of the REAL ``ManagerBasedRLEnv._reset_idx`` + EventManager (reset terms with
    ``min_step_count_between_reset``, per-env and global interval timers) + CommandManager + CurriculumManager / terrain_levels_vel:
    what the terms write to the simulator, trigger state, timers, commands and metrics, terrain levels / origins, log entries."""
    from _util import OrchGolden
    from oracle.orchestration_oracle import OrchestrationOracle

    g = OrchGolden()
    N, meta = g.N, g.meta
    orc = OrchestrationOracle(g.fixture["env"], N, g.robot.num_joints, g.robot.body_names, meta["step_dt"], meta["max_episode_length_s"],
                              g.t("static/default_root_state"), g.t("static/default_joint_pos"), g.t("static/default_joint_vel"),
                              g.t("static/soft_joint_pos_limits"), g.t("static/soft_joint_vel_limits"), g.t("terrain/origins"),
                              g.t("terrain/levels0"), g.t("terrain/types"), meta["terrain"]["size_x"], g.t("interval/time_left_init"))
    assert [n for n, _ in orc.terms] == g.term_names

    def feed(tag):
        return {k: g.t(f"{tag}/in/{k}") for k in ("root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w")}

    def check(tag):
        for k, v in orc.sim_writes.items():
            assert_close(v, g.t(f"{tag}/sim_writes/{k}"), 1e-5, f"{tag} sim_writes[{k}]")
        assert torch.equal(orc.levels, g.t(f"{tag}/terrain_levels")) and torch.equal(orc.env_origins, g.t(f"{tag}/env_origins")), tag
        c = orc.cmd
        for k, a in (("command", c.vel_command_b), ("command_time_left", c.time_left), ("heading_target", c.heading_target),
                     ("metric_error_vel_xy", c.metrics["error_vel_xy"]), ("metric_error_vel_yaw", c.metrics["error_vel_yaw"])):
            assert_close(a, g.t(f"{tag}/{k}"), 1e-5, f"{tag} {k}")
        assert torch.equal(c.command_counter, g.t(f"{tag}/command_counter")) and torch.equal(c.is_standing_env, g.t(f"{tag}/is_standing_env"))
        tl = torch.stack([orc.time_left[n].expand(N) for n in g.interval_names])
        assert_close(tl, g.t(f"{tag}/interval_time_left"), 1e-6, f"{tag} interval timers")
        assert torch.equal(torch.stack([orc.last_triggered[n] for n in g.reset_names]), g.t(f"{tag}/reset_last_triggered_step")), tag
        assert torch.equal(torch.stack([orc.triggered_once[n] for n in g.reset_names]), g.t(f"{tag}/reset_triggered_once")), tag
        ref_log = g.log(tag)
        for k, v in orc.log.items():
            assert abs(ref_log[k] - v) <= 1e-5 * max(1.0, abs(v)), (tag, k, ref_log[k], v)

    d = g.draws(0)
    orc.cmd._draw[:] = 0
    orc.reset_idx(torch.arange(N), feed("reset"), 0, d, d["command"], d["rand_levels"])
    check("reset")
    pushes = resets = 0
    for s in range(g.steps):
        tag, d = f"step{s}", g.draws(s + 1)
        ids, f = g.t(f"{tag}/reset_env_ids"), feed(tag)
        orc.cmd._draw[:] = 0
        if len(ids) > 0:
            orc.reset_idx(ids, f, s + 1, d, d["command"], d["rand_levels"])
            resets += len(ids)
        fired = orc.step_tail(f, d, d["interval"], d["command"])
        pushes += sum(len(v) for v in fired.values())
        check(tag)
    assert resets == meta["n_resets"] and pushes == meta["n_push"] + meta["n_global_push"] * N  # (a global timer pushes every env at once)
