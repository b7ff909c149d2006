"""SURVEY 8f row 1 (ContactSensor update, UniformVelocityCommand): oracle vs the fixtures produced by the real reference
classes (CPU), HIP vs the same fixtures (GPU)."""

import json
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, assert_close

Z = np.load(os.path.join(GOLDEN, "producers.npz"))


def t(key):
    return torch.from_numpy(np.ascontiguousarray(Z[key]))


CONTACT_KEYS = ("net_forces_w", "net_forces_w_history", "last_air_time", "current_air_time", "last_contact_time",
                "current_contact_time")
CMD_KEYS = ("vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "time_left", "command_counter")


def test_contact_sensor_oracle_matches_reference():
    from oracle.producers_oracle import contact_sensor_update

    m = json.loads(str(Z["contact/meta"]))
    N, B, H = m["N"], m["B"], m["H"]
    st = dict(timestamp=torch.zeros(N), timestamp_last_update=torch.zeros(N), is_outdated=torch.ones(N, dtype=torch.bool),
              net_forces_w=torch.zeros(N, B, 3), net_forces_w_history=torch.zeros(N, H, B, 3), last_air_time=torch.zeros(N, B),
              current_air_time=torch.zeros(N, B), last_contact_time=torch.zeros(N, B), current_contact_time=torch.zeros(N, B))
    forces = t("contact/forces")
    for k in range(m["steps"]):
        if k == m["reset_step"]:
            ids = torch.tensor(m["reset_ids"])
            for name in ("timestamp", "timestamp_last_update", "net_forces_w", "net_forces_w_history", "current_air_time",
                         "last_air_time", "current_contact_time", "last_contact_time"):  # contact_sensor.py:143-170
                st[name][ids] = 0.0
            st["is_outdated"][ids] = True
        contact_sensor_update(st, forces[k], m["dt"], m["update_period"], m["force_threshold"], H, True)
        for name in CONTACT_KEYS:
            assert torch.equal(st[name], t(f"contact/step{k}/{name}")), (k, name)
        assert torch.equal(st["timestamp"], t(f"contact/step{k}/timestamp"))


def test_velocity_command_oracle_matches_reference():
    from oracle.producers_oracle import VelocityCommandOracle

    m = json.loads(str(Z["command/meta"]))
    orc = VelocityCommandOracle(m["cfg"], m["N"], m["step_dt"])
    for k in range(m["steps"]):
        tag = f"command/step{k}"
        orc.reset_and_compute(m["step_dt"], t(f"{tag}/root_quat_w"), t(f"{tag}/root_lin_vel_w"), t(f"{tag}/root_ang_vel_w"),
                              t(f"{tag}/reset_mask"), t(f"{tag}/uniforms"))
        for name in CMD_KEYS:
            got = getattr(orc, name)
            ref = t(f"{tag}/{name}")
            if got.dtype in (torch.bool, torch.long):
                assert torch.equal(got, ref), (k, name)
            else:
                assert_close(got, ref, 1e-6, f"step {k} {name}")
        assert_close(orc.metrics["error_vel_xy"], t(f"{tag}/error_vel_xy"), 1e-6, "error_vel_xy")
        assert_close(orc.metrics["error_vel_yaw"], t(f"{tag}/error_vel_yaw"), 1e-6, "error_vel_yaw")


@pytest.mark.gpu
def test_contact_sensor_hip_matches_reference():
    from isaaclab_amd.producers import ContactSensorState

    m = json.loads(str(Z["contact/meta"]))
    s = ContactSensorState(m["N"], m["B"], m["H"], True, m["update_period"], m["force_threshold"], "cuda:0")
    forces = t("contact/forces").cuda()
    for k in range(m["steps"]):
        if k == m["reset_step"]:
            s.reset(torch.tensor(m["reset_ids"], device="cuda"))
        s.update(forces[k], m["dt"])
        for name in CONTACT_KEYS:
            assert torch.equal(getattr(s.data, name).cpu(), t(f"contact/step{k}/{name}")), (k, name)
        assert torch.equal(s._timestamp.cpu(), t(f"contact/step{k}/timestamp"))
        assert torch.equal(s._timestamp_last_update.cpu(), t(f"contact/step{k}/timestamp_last_update"))
        assert torch.equal(s.compute_first_contact(0.02).cpu(), t(f"contact/step{k}/first_contact"))


@pytest.mark.gpu
def test_velocity_command_hip_matches_reference():
    from isaaclab_amd.producers import UniformVelocityCommand

    m = json.loads(str(Z["command/meta"]))
    cmd = UniformVelocityCommand(m["cfg"], m["N"], m["step_dt"], "cuda:0")
    for k in range(m["steps"]):
        tag = f"command/step{k}"
        cmd.compute(m["step_dt"], t(f"{tag}/root_quat_w").cuda(), t(f"{tag}/root_lin_vel_w").cuda(),
                    t(f"{tag}/root_ang_vel_w").cuda(), t(f"{tag}/reset_mask").cuda(), t(f"{tag}/uniforms").cuda())
        for name in CMD_KEYS:
            got, ref = getattr(cmd, name).cpu(), t(f"{tag}/{name}")
            if ref.dtype in (torch.bool, torch.long):
                assert torch.equal(got, ref), (k, name)
            else:
                assert_close(got, ref, 1e-5, f"step {k} {name}")
        assert_close(cmd.metrics["error_vel_xy"], t(f"{tag}/error_vel_xy"), 1e-5, "error_vel_xy")
        assert_close(cmd.metrics["error_vel_yaw"], t(f"{tag}/error_vel_yaw"), 1e-5, "error_vel_yaw")
    # in-kernel generator: ranges respected, standing envs zeroed
    cmd.compute(m["step_dt"], t("command/step0/root_quat_w").cuda(), t("command/step0/root_lin_vel_w").cuda(),
                t("command/step0/root_ang_vel_w").cuda(), torch.ones(m["N"], dtype=torch.bool, device="cuda"), None)
    c = cmd.command.cpu()
    assert float(c.abs().max()) <= 1.0 + 1e-6 and bool((c[cmd.is_standing_env.cpu()] == 0).all())


def test_articulation_oracle_matches_reference():
    from oracle.mdp_oracle import convert_quat

    m = json.loads(str(Z["artic/meta"]))
    prev = t("artic/initial_joint_vel").clone()
    sim_t, acc_t = 0.0, -1.0  # TimestampedBuffer starts at -1.0: the first difference divides by dt + 1.0
    for k in range(m["steps"]):
        tag = f"artic/step{k}"
        sim_t += m["dt"]
        elapsed, acc_t = sim_t - acc_t, sim_t
        tf, vel, dv = t(f"{tag}/root_transforms"), t(f"{tag}/root_velocities"), t(f"{tag}/dof_velocities")
        # articulation_data.py:365-380, 546-556
        assert torch.equal(tf[:, :3], t(f"{tag}/root_pos_w"))
        assert torch.equal(convert_quat(tf[:, 3:7], to="wxyz"), t(f"{tag}/root_quat_w"))
        assert torch.equal(vel[:, :3], t(f"{tag}/root_lin_vel_w")) and torch.equal(vel[:, 3:], t(f"{tag}/root_ang_vel_w"))
        assert_close((dv - prev) / elapsed, t(f"{tag}/joint_acc"), 1e-6, "joint_acc")
        prev = dv.clone()


@pytest.mark.gpu
def test_articulation_hip_matches_reference():
    from isaaclab_amd.producers import ArticulationRootState

    m = json.loads(str(Z["artic/meta"]))
    st = ArticulationRootState(m["N"], m["J"], "cuda:0", t("artic/initial_joint_vel").cuda())
    for k in range(m["steps"]):
        tag = f"artic/step{k}"
        st.update(t(f"{tag}/root_transforms").cuda(), t(f"{tag}/root_velocities").cuda(), t(f"{tag}/dof_velocities").cuda(), m["dt"])
        for name in ("root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w"):
            assert torch.equal(getattr(st, name).cpu(), t(f"{tag}/{name}")), (k, name)
        assert_close(st.joint_acc, t(f"{tag}/joint_acc"), 1e-5, "joint_acc")


@pytest.mark.gpu
def test_empirical_normalization_matches_rsl_rl_restatement():
    from isaaclab_amd.rsl_rl.normalizer import EmpiricalNormalization
    from oracle.rsl_rl_oracle import EmpiricalNormalizationOracle

    D = 235
    norm = EmpiricalNormalization([D]).cuda()
    orc = EmpiricalNormalizationOracle(D)
    g = torch.Generator().manual_seed(5)
    for k in range(4):
        x = torch.randn(4096 if k else 64, D, generator=g) * (1 + k) + 0.3 * k
        got = norm(x.cuda())
        ref = orc.forward(x, training=True)
        assert_close(got, ref, 1e-5, f"normalised obs, batch {k}")
        assert_close(norm._mean, orc.mean, 1e-5, "mean") and assert_close(norm._var, orc.var, 1e-5, "var") is None
    assert norm.count == orc.count
    norm.eval()
    x = torch.randn(10, D, generator=g)
    m0 = norm._mean.clone()
    assert_close(norm(x.cuda()), orc.forward(x, training=False), 1e-5, "eval mode")
    assert torch.equal(norm._mean, m0)


# ------------------------------------------------------------------------------------------------ reset events (8f row 2)
def _events_golden():
    import json

    z = np.load(os.path.join(GOLDEN, "events.npz"))
    return z, json.loads(str(z["meta"]))


@pytest.mark.gpu
def test_reset_events_hip_matches_reference():
    """imx_reset_events / imx_push_velocity fed the reference's uniform draws vs what the REAL reference terms wrote."""
    from isaaclab_amd.events import ResetEvents

    z, meta = _events_golden()
    N, J = meta["N"], meta["J"]
    c = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    mask = c("mask").to(torch.uint8)
    mb = c("mask")
    for tag, mode in (("scale", "scale"), ("scale2", "scale"), ("offset", "offset")):
        r = z[f"joints_{tag}/ranges"]
        ev = ResetEvents(N, J, "cuda", meta["pose_range"], meta["velocity_range"], (float(r[0]), float(r[1])), (float(r[2]), float(r[3])),
                         mode, meta["push_range"])
        U = torch.cat([c("root/u_pose"), c("root/u_vel"), c(f"joints_{tag}/u_pos"), c(f"joints_{tag}/u_vel")], dim=1).contiguous()
        pose, vel = torch.full((N, 7), 7.0, device="cuda"), torch.full((N, 6), 7.0, device="cuda")
        jp, jv = torch.full((N, J), 7.0, device="cuda"), torch.full((N, J), 7.0, device="cuda")
        ev.reset(mask, c("default_root_state"), c("env_origins"), pose, vel, c("default_joint_pos"), c("default_joint_vel"),
                 c("soft_joint_pos_limits"), c("soft_joint_vel_limits"), jp, jv, uniforms=U)
        assert_close(pose[mb], c("root/pose_out")[mb], 1e-6, "root pose")
        assert_close(vel[mb], c("root/vel_out")[mb], 1e-6, "root velocity")
        assert_close(jp[mb], c(f"joints_{tag}/pos_out")[mb], 1e-6, f"joint pos {tag}")
        assert_close(jv[mb], c(f"joints_{tag}/vel_out")[mb], 1e-6, f"joint vel {tag}")
        for t in (pose, vel, jp, jv):  # rows of envs that did not reset are untouched
            assert bool((t[~mb] == 7.0).all())
    v = c("root_vel_w").clone()
    ev.push(mask, v, uniforms=c("push/u"))
    assert_close(v[mb], c("push/vel_out")[mb], 1e-6, "push")
    assert torch.equal(v[~mb], c("root_vel_w")[~mb])


@pytest.mark.gpu
def test_reset_events_in_kernel_generator_properties():
    """Performance mode (counter-based generator): samples inside the ranges, deterministic per (seed, step), fresh per step."""
    from isaaclab_amd.events import ResetEvents

    N, J = 4096, 12
    g = torch.Generator().manual_seed(5)
    drs = torch.zeros(N, 13)
    drs[:, 3] = 1.0
    drs[:, 2] = 0.6
    drs, org = drs.cuda(), (torch.randn(N, 3, generator=g) * 5).cuda()
    djp = (torch.rand(N, J, generator=g) + 0.5).cuda()
    djv = torch.ones(N, J).cuda()
    plim = torch.stack([torch.full((N, J), -10.0), torch.full((N, J), 10.0)], dim=-1).cuda()
    vlim = torch.full((N, J), 0.5).cuda()
    cfg = {"reset_base": {"func": "isaaclab.envs.mdp.events:reset_root_state_uniform", "mode": "reset",
                          "params": {"pose_range": {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "yaw": (-3.14, 3.14)},
                                     "velocity_range": {"x": (-0.5, 0.5), "roll": (-0.2, 0.2)}}},
           "reset_robot_joints": {"func": "isaaclab.envs.mdp.events:reset_joints_by_scale", "mode": "reset",
                                  "params": {"position_range": (0.5, 1.5), "velocity_range": (-2.0, 2.0)}},
           "push_robot": {"func": "isaaclab.envs.mdp.events:push_by_setting_velocity", "mode": "interval",
                          "params": {"velocity_range": {"x": (-0.5, 0.5), "y": (-0.5, 0.5)}}}}
    outs = []
    for trial in range(2):
        ev = ResetEvents.from_cfg(cfg, N, J, "cuda", seed=9)
        pose, vel, jp, jv = (torch.zeros(N, 7, device="cuda"), torch.zeros(N, 6, device="cuda"), torch.zeros(N, J, device="cuda"),
                             torch.zeros(N, J, device="cuda"))
        ev.reset(None, drs, org, pose, vel, djp, djv, plim, vlim, jp, jv)
        first = pose.clone()
        ev.reset(None, drs, org, pose, vel, djp, djv, plim, vlim, jp, jv)
        outs.append((first, pose.clone(), vel.clone(), jp.clone(), jv.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))  # same seed, same step sequence -> same samples
    first, pose, vel, jp, jv = outs[0]
    assert not torch.equal(first, pose)  # the step counter advanced
    d = pose[:, :3] - org - drs[:, :3]
    assert float(d[:, :2].abs().max()) <= 0.5 + 1e-6 and float(d[:, 2].abs().max()) <= 1e-6
    assert 0.2 < float(d[:, 0].std()) < 0.35  # U(-0.5, 0.5): sigma = 0.289
    assert_close(pose[:, 3:].norm(dim=1), torch.ones(N), 1e-5, "unit quaternion")
    assert float(pose[:, 4:6].abs().max()) <= 1e-6  # yaw only
    assert float(vel[:, 0].abs().max()) <= 0.5 and float(vel[:, 3].abs().max()) <= 0.2 and float(vel[:, [1, 2, 4, 5]].abs().max()) == 0.0
    ratio = jp / djp
    assert float(ratio.min()) >= 0.5 - 1e-6 and float(ratio.max()) <= 1.5 + 1e-6
    assert float(jv.abs().max()) <= 0.5  # clamped to the soft velocity limit
    v = torch.zeros(N, 6, device="cuda")
    ev.push(None, v)
    assert float(v[:, :2].abs().max()) <= 0.5 and float(v[:, 2:].abs().max()) == 0.0 and float(v[:, 0].std()) > 0.2


@pytest.mark.gpu
def test_external_force_torque_hip_matches_reference():
    from isaaclab_amd.events import ExternalForceTorque

    z, meta = _events_golden()
    N = meta["N"]
    c = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    r = z["ext/ranges"]
    ids = z["ext/body_ids"].tolist()
    ev = ExternalForceTorque(N, 17, "cuda", (float(r[0]), float(r[1])), (float(r[2]), float(r[3])), body_ids=ids)
    forces, torques = torch.full((N, 17, 3), 9.0, device="cuda"), torch.full((N, 17, 3), 9.0, device="cuda")
    U = torch.stack([c("ext/u_force"), c("ext/u_torque")]).contiguous()
    mb = c("mask")
    ev.apply(mb.to(torch.uint8), forces, torques, uniforms=U)
    assert_close(forces[mb][:, ids], c("ext/forces")[mb], 1e-6, "forces")
    assert_close(torques[mb][:, ids], c("ext/torques")[mb], 1e-6, "torques")
    other = [b for b in range(17) if b not in ids]
    assert bool((forces[:, other] == 9.0).all()) and bool((forces[~mb] == 9.0).all()) and bool((torques[~mb] == 9.0).all())
    ev.apply(None, forces, torques)  # in-kernel generator, every env
    assert float(forces[:, ids].min()) >= r[0] and float(forces[:, ids].max()) <= r[1] and float(forces[:, ids].std()) > 1.0
    assert float(torques[:, ids].min()) >= r[2] and float(torques[:, ids].max()) <= r[3]


@pytest.mark.gpu
def test_terrain_curriculum_hip_matches_reference():
    from isaaclab_amd.events import TerrainCurriculum

    z, meta = _events_golden()
    c = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    levels, origins = c("curr/levels_in").clone(), c("curr/env_origins_in").clone()
    cur = TerrainCurriculum(c("curr/terrain_origins"), levels, c("curr/types"), origins, meta["terrain_size"], meta["max_episode_length_s"])
    mean = cur.update(c("mask").to(torch.uint8), c("curr/root_pos_w"), c("curr/command"), rand_levels=c("curr/randint"))
    assert torch.equal(levels, c("curr/levels_out"))
    assert torch.equal(origins, c("curr/env_origins_out"))
    assert abs(float(mean) - float(z["curr/mean_level"])) < 1e-6
    # in-kernel draw for solved top levels: stays inside [0, R)
    levels2 = torch.full_like(levels, meta["R"] - 1)
    far = c("curr/env_origins_in").clone()
    cur2 = TerrainCurriculum(c("curr/terrain_origins"), levels2, c("curr/types"), far.clone(), meta["terrain_size"], meta["max_episode_length_s"])
    pos = far.clone()
    pos[:, 0] += 5.0  # walked 5 m > half a tile: everyone moves up from the last level
    cur2.update(None, pos, c("curr/command"))
    assert int(levels2.min()) >= 0 and int(levels2.max()) < meta["R"] and len(torch.unique(levels2)) > 3


# ------------------------------------------------------------------------------------------------ actuators (8f row 4)
def _act_golden():
    return np.load(os.path.join(GOLDEN, "actuators.npz"))


def test_actuator_oracle_matches_reference():
    from oracle.producers_oracle import actuator_pd

    z = _act_golden()
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    sat = json.loads(str(z["meta"]))["saturation_effort"]
    for tag, kw in (("ideal", {}), ("dc", dict(velocity_limit=t("velocity_limit"), saturation_effort=sat))):
        c, a = actuator_pd(t(f"{tag}/q_des"), t(f"{tag}/qd_des"), t(f"{tag}/ff"), t(f"{tag}/q"), t(f"{tag}/qd"), t("stiffness"), t("damping"),
                           t("effort_limit"), **kw)
        assert_close(c, t(f"{tag}/computed"), 1e-6, f"{tag} computed")
        assert_close(a, t(f"{tag}/applied"), 1e-6, f"{tag} applied")


@pytest.mark.gpu
def test_actuator_hip_matches_reference():
    from isaaclab_amd.producers import PDActuator

    z = _act_golden()
    c = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    sat = json.loads(str(z["meta"]))["saturation_effort"]
    for tag, kw in (("ideal", {}), ("dc", dict(velocity_limit=c("velocity_limit"), saturation_effort=sat))):
        act = PDActuator(c("stiffness"), c("damping"), c("effort_limit"), **kw)
        applied = act.compute(c(f"{tag}/q_des"), c(f"{tag}/q"), c(f"{tag}/qd"), c(f"{tag}/qd_des"), c(f"{tag}/ff"))
        assert_close(act.computed_effort, c(f"{tag}/computed"), 1e-5, f"{tag} computed")
        assert_close(applied, c(f"{tag}/applied"), 1e-5, f"{tag} applied")
        assert float((applied != act.computed_effort).float().mean()) > 0.05  # the limits are exercised


def _delayed_steps(z, tag):
    meta = json.loads(str(z["meta"]))
    for t in range(meta["delayed_steps"]):
        pre = f"{tag}/step{t}/"
        yield t, pre, (z[pre + "reset_ids"] if pre + "reset_ids" in z.files else None)


def test_delayed_and_remotized_actuator_oracle_matches_reference():
    """DelayedPDActuator / RemotizedPDActuator (actuator_pd.py:289-412) restated, against 9 steps of the real classes with
    partial resets and re-drawn lags."""
    from oracle.producers_oracle import DelayedPDOracle

    z = _act_golden()
    t_ = lambda k: torch.from_numpy(z[k])  # noqa: E731
    meta = json.loads(str(z["meta"]))
    for tag in ("delayed", "remotized"):
        o = DelayedPDOracle(meta["N"], meta["J"], meta["max_delay"], t_("stiffness"), t_("damping"), t_("effort_limit"),
                            t_("remotized/lookup") if tag == "remotized" else None)
        changed = 0
        for t, pre, ids in _delayed_steps(z, tag):
            if ids is not None:
                o.reset(torch.from_numpy(ids), t_(pre + "time_lags"))
            c, a = o.compute(t_(pre + "q_des"), t_(pre + "qd_des"), t_(pre + "ff"), t_(pre + "q"), t_(pre + "qd"))
            assert_close(c, t_(pre + "computed"), 1e-6, f"{tag} step {t} computed")
            assert_close(a, t_(pre + "applied"), 1e-6, f"{tag} step {t} applied")
            undelayed = t_("stiffness") * (t_(pre + "q_des") - t_(pre + "q")) + t_("damping") * (t_(pre + "qd_des") - t_(pre + "qd")) + t_(pre + "ff")
            changed += int((undelayed - c).abs().max() > 1e-3)
        assert changed >= 6  # the delay line is exercised


@pytest.mark.gpu
def test_delayed_and_remotized_actuator_hip_matches_reference():
    from isaaclab_amd.producers import DelayedPDActuator

    z = _act_golden()
    c_ = lambda k: torch.from_numpy(z[k]).cuda()  # noqa: E731
    meta = json.loads(str(z["meta"]))
    for tag in ("delayed", "remotized"):
        act = DelayedPDActuator(c_("stiffness"), c_("damping"), 0, meta["max_delay"], effort_limit=c_("effort_limit"),
                                joint_parameter_lookup=z["remotized/lookup"] if tag == "remotized" else None)
        clipped = 0.0
        for t, pre, ids in _delayed_steps(z, tag):
            if ids is not None:
                act.reset(torch.from_numpy(ids).cuda(), c_(pre + "time_lags"))
            applied = act.compute(c_(pre + "q_des"), c_(pre + "q"), c_(pre + "qd"), c_(pre + "qd_des"), c_(pre + "ff"))
            assert_close(act.computed_effort, c_(pre + "computed"), 1e-5, f"{tag} step {t} computed")
            assert_close(applied, c_(pre + "applied"), 1e-5, f"{tag} step {t} applied")
            clipped += float((applied != act.computed_effort).float().mean())
        assert clipped / meta["delayed_steps"] > 0.05
    # a drawn lag stays inside [min_delay, max_delay]
    act = DelayedPDActuator(c_("stiffness"), c_("damping"), 1, 3)
    act.reset()
    assert int(act.time_lags.min()) >= 1 and int(act.time_lags.max()) <= 3
    with pytest.raises(ValueError):
        DelayedPDActuator(c_("stiffness"), c_("damping"), 3, 1)


# ---- ActuatorNetLSTM / ActuatorNetMLP (the ANYdrive-3 LSTM is the actuator of the Anymal-C task robot): fixture from
# the REAL classes driving synthetic TorchScript networks (oracle/gen_golden_actuator_nets.py)
ZN = np.load(os.path.join(GOLDEN, "actuator_nets.npz"))


def tn(key):
    return torch.from_numpy(np.ascontiguousarray(ZN[key]))


def _net_layers():
    lstm = [(tn(f"lstm/net/lstm.weight_ih_l{k}"), tn(f"lstm/net/lstm.weight_hh_l{k}"), tn(f"lstm/net/lstm.bias_ih_l{k}"),
             tn(f"lstm/net/lstm.bias_hh_l{k}")) for k in range(2)]
    head = [(tn("lstm/net/fc1.weight"), tn("lstm/net/fc1.bias")), (tn("lstm/net/fc2.weight"), tn("lstm/net/fc2.bias"))]
    mlp = [(tn(f"mlp/net/{k}.weight"), tn(f"mlp/net/{k}.bias")) for k in (0, 2, 4)]
    return lstm, head, mlp


def _drive_nets(make, tol):
    m = json.loads(str(ZN["meta"]))
    lstm, head, mlp = _net_layers()
    for tag in ("lstm", "mlp"):
        a, dev = make(tag, m, lstm, head, mlp)
        for k in range(m["steps"]):
            if f"{tag}/step{k}/reset_ids" in ZN:
                a.reset(tn(f"{tag}/step{k}/reset_ids").to(dev))
            out = a.compute(tn(f"{tag}/step{k}/q_des").to(dev), tn(f"{tag}/step{k}/q").to(dev), tn(f"{tag}/step{k}/qd").to(dev))
            computed, applied = out if isinstance(out, tuple) else (a.computed_effort, a.applied_effort)
            assert_close(computed, tn(f"{tag}/step{k}/computed"), tol, f"{tag} step {k} computed effort")
            assert_close(applied, tn(f"{tag}/step{k}/applied"), tol, f"{tag} step {k} applied effort")
        yield tag, a


def test_actuator_net_oracle_matches_reference():
    from oracle.producers_oracle import ActuatorNetLSTMOracle, ActuatorNetMLPOracle

    def make(tag, m, lstm, head, mlp):
        if tag == "lstm":
            return ActuatorNetLSTMOracle(m["N"], m["J"], lstm, head, "softsign", m["saturation_effort"], m["effort_limit"], m["velocity_limit"]), "cpu"
        c = m["mlp"]
        return ActuatorNetMLPOracle(m["N"], m["J"], mlp, c["activation"], c["input_idx"], c["pos_scale"], c["vel_scale"], c["torque_scale"],
                                    c["input_order"], m["saturation_effort"], m["effort_limit"], m["velocity_limit"]), "cpu"

    # (1e-5: torch's fused LSTM cell and this gate-by-gate restatement round the 10- and 16-term dot products in different orders)
    for tag, a in _drive_nets(make, 1e-5):
        if tag == "lstm":
            assert_close(a.h, tn("lstm/final_hidden"), 1e-5, "hidden state")
            assert_close(a.c, tn("lstm/final_cell"), 1e-5, "cell state")
        else:
            assert_close(a.pos_hist, tn("mlp/final_pos_hist"), 0.0, "position-error history")


@pytest.mark.gpu
def test_actuator_net_hip_matches_reference():
    """One launch per compute(): LSTM stack + head (or the history MLP) + DC-motor clip for every (env, joint) sample, 8 steps with
    partial resets, against what the real ActuatorNetLSTM / ActuatorNetMLP computed with the same networks."""
    from isaaclab_amd.producers import ActuatorNetLSTM, ActuatorNetMLP

    def make(tag, m, lstm, head, mlp):
        if tag == "lstm":
            return ActuatorNetLSTM(m["N"], m["J"], m["effort_limit"], m["velocity_limit"], m["saturation_effort"], lstm_layers=lstm, head=head,
                                   head_activation="softsign"), "cuda:0"
        c = m["mlp"]
        return ActuatorNetMLP(m["N"], m["J"], m["effort_limit"], m["velocity_limit"], m["saturation_effort"], c["input_idx"], c["pos_scale"],
                              c["vel_scale"], c["torque_scale"], c["input_order"], layers=mlp, activation=c["activation"]), "cuda:0"

    for tag, a in _drive_nets(make, 1e-5):
        if tag == "lstm":
            assert_close(a.sea_hidden_state, tn("lstm/final_hidden"), 1e-5, "hidden state")
            assert_close(a.sea_cell_state, tn("lstm/final_cell"), 1e-5, "cell state")
        else:
            assert_close(a._joint_pos_error_history, tn("mlp/final_pos_hist"), 0.0, "position-error history")
            assert_close(a._joint_vel_history, tn("mlp/final_vel_hist"), 0.0, "velocity history")
    # the TorchScript route of the constructor: layers read out of a scripted module like the reference's
    import torch.nn as nn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.lstm = nn.LSTM(2, 8, 2, batch_first=True)
            self.fc1, self.act, self.fc2 = nn.Linear(8, 16), nn.Softsign(), nn.Linear(16, 1)

        def forward(self, x: torch.Tensor, hc: tuple[torch.Tensor, torch.Tensor]):
            y, (h, c) = self.lstm(x, hc)
            return self.fc2(self.act(self.fc1(y[:, -1]))), (h, c)

    torch.manual_seed(3)
    net = torch.jit.script(Net())
    a = ActuatorNetLSTM(16, 12, 80.0, 7.5, 120.0, network=net)
    q_des, q, qd = (torch.randn(16, 12, device="cuda:0") for _ in range(3))
    got = a.compute(q_des, q, qd).cpu()
    x = torch.stack([(q_des - q).flatten(), qd.flatten()], 1).cpu().unsqueeze(1)
    ref, _ = net(x, (torch.zeros(2, 192, 8), torch.zeros(2, 192, 8)))
    assert_close(a.computed_effort, ref.reshape(16, 12), 1e-5, "scripted module")
    assert torch.isfinite(got).all()
