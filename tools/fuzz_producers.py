"""Random sweeps of two stateful producers against their oracles (which the reference's fixtures pin): ContactSensor update (random
body / history counts, update period gating, thresholds, forces, partial resets) and UniformVelocityCommand (random ranges, heading /
standing fractions, resampling windows shorter and longer than a step, resets, fed uniforms).  Test infrastructure, run on the GPU box:
    python tools/fuzz_producers.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from _util import assert_close

CONTACT_KEYS = ("net_forces_w", "net_forces_w_history", "last_air_time", "current_air_time", "last_contact_time", "current_contact_time")
CMD_KEYS = ("vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "time_left", "command_counter")


def case_contact(rng):
    from isaaclab_amd.producers import ContactSensorState
    from oracle.producers_oracle import contact_sensor_update

    N, B, H = int(rng.choice([1, 7, 64, 65, 1000, 4096])), int(rng.choice([1, 4, 17, 30])), int(rng.choice([0, 1, 3, 5]))
    dt = float(rng.choice([0.005, 0.02]))
    period = float(rng.choice([0.0, dt, 2 * dt, 0.05]))
    thr = float(rng.choice([1.0, 0.1, 5.0]))
    steps = int(rng.integers(3, 10))
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    s = ContactSensorState(N, B, H, True, period, thr, "cuda:0")
    st = dict(timestamp=torch.zeros(N), timestamp_last_update=torch.zeros(N), is_outdated=torch.ones(N, dtype=torch.bool),
              net_forces_w=torch.zeros(N, B, 3), net_forces_w_history=torch.zeros(N, max(H, 0), B, 3), last_air_time=torch.zeros(N, B),
              current_air_time=torch.zeros(N, B), last_contact_time=torch.zeros(N, B), current_contact_time=torch.zeros(N, B))
    for k in range(steps):
        if rng.random() < 0.4:
            ids = torch.nonzero(torch.rand(N, generator=g) < 0.3).flatten()
            if len(ids):
                for name in ("timestamp", "timestamp_last_update", "net_forces_w", "net_forces_w_history", "current_air_time", "last_air_time",
                             "current_contact_time", "last_contact_time"):
                    st[name][ids] = 0.0
                st["is_outdated"][ids] = True
                s.reset(ids.cuda())
        f = torch.randn(N, B, 3, generator=g) * (torch.rand(N, B, 1, generator=g) < 0.5) * float(rng.choice([0.5, 3.0, 20.0]))
        contact_sensor_update(st, f, dt, period, thr, H, True)
        s.update(f.cuda(), dt)
        for name in CONTACT_KEYS:
            if name == "net_forces_w_history" and H == 0:
                continue
            assert torch.equal(getattr(s.data, name).cpu(), st[name]), (k, name)
        assert torch.equal(s._timestamp.cpu(), st["timestamp"]) and torch.equal(s._timestamp_last_update.cpu(), st["timestamp_last_update"]), (k, "stamps")
    return f"N={N} B={B} H={H} dt={dt} period={period} thr={thr} steps={steps}"


def case_command(rng):
    from isaaclab_amd.producers import UniformVelocityCommand
    from oracle.producers_oracle import VelocityCommandOracle

    N, step_dt = int(rng.choice([1, 63, 64, 65, 1000, 4096])), float(rng.choice([0.02, 0.005]))
    lo = float(rng.choice([0.5, 2.0, 10.0])) * step_dt
    rng2 = lambda a: [-float(a), float(a)]  # noqa: E731
    cfg = {"resampling_time_range": [lo, lo * float(rng.choice([1.0, 1.5, 3.0]))], "heading_command": bool(rng.integers(0, 2)),
           "heading_control_stiffness": float(rng.choice([0.5, 1.0])), "rel_standing_envs": float(rng.choice([0.0, 0.2, 1.0])),
           "rel_heading_envs": float(rng.choice([0.0, 0.7, 1.0])),
           "ranges": {"lin_vel_x": rng2(rng.choice([1.0, 0.3])), "lin_vel_y": [0.0, float(rng.choice([0.0, 0.5]))], "ang_vel_z": rng2(rng.choice([1.0, 2.0])),
                      "heading": [-3.141592653589793, 3.141592653589793]}}
    steps = int(rng.integers(3, 12))
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    orc = VelocityCommandOracle(cfg, N, step_dt)
    cmd = UniformVelocityCommand(cfg, N, step_dt, "cuda:0")
    for k in range(steps):
        q = torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)
        lin, ang = torch.randn(N, 3, generator=g), torch.randn(N, 3, generator=g)
        mask = torch.rand(N, generator=g) < float(rng.choice([0.0, 0.1, 1.0]))
        U = torch.rand(2, N, 7, generator=g)
        orc.reset_and_compute(step_dt, q, lin, ang, mask, U)
        cmd.compute(step_dt, q.cuda(), lin.cuda(), ang.cuda(), mask.cuda(), U.cuda())
        for name in CMD_KEYS:
            got, ref = getattr(cmd, name).cpu(), getattr(orc, name)
            if ref.dtype in (torch.bool, torch.long):
                assert torch.equal(got, ref), (k, name)
            else:
                assert_close(got, ref, 1e-5, f"step {k} {name}")
        assert_close(cmd.metrics["error_vel_xy"], orc.metrics["error_vel_xy"], 1e-5, "error_vel_xy")
        assert_close(cmd.metrics["error_vel_yaw"], orc.metrics["error_vel_yaw"], 1e-5, "error_vel_yaw")
    return f"N={N} step_dt={step_dt} resample={cfg['resampling_time_range']} heading={cfg['heading_command']} steps={steps}"


def case_pd_actuator(rng):
    from isaaclab_amd.producers import PDActuator
    from oracle.producers_oracle import actuator_pd

    N, J = int(rng.choice([1, 7, 64, 1000, 4096])), int(rng.choice([1, 2, 12, 23, 37]))
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    r = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    stiff, damp, elim, vlim = 20 + 80 * r(N, J), 0.5 + 4 * r(N, J), 20 + 60 * r(N, J), 2 + 8 * r(N, J)
    sat = float(rng.choice([60.0, 120.0]))
    q_des, q, qd, qd_des, ff = (torch.randn(N, J, generator=g) * float(rng.choice([0.3, 2.0])) for _ in range(5))
    dc = bool(rng.integers(0, 2))
    kw = dict(velocity_limit=vlim, saturation_effort=sat) if dc else {}
    c0, a0 = actuator_pd(q_des, qd_des, ff, q, qd, stiff, damp, elim, **kw)
    act = PDActuator(stiff.cuda(), damp.cuda(), elim.cuda(), **({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}))
    applied = act.compute(q_des.cuda(), q.cuda(), qd.cuda(), qd_des.cuda(), ff.cuda())
    assert_close(act.computed_effort, c0, 1e-5, "computed effort")
    assert_close(applied, a0, 1e-5, "applied effort")
    return f"N={N} J={J} dc_motor={dc}"


def case_articulation(rng):
    from isaaclab_amd.producers import ArticulationRootState
    from oracle.mdp_oracle import convert_quat

    N, J, dt = int(rng.choice([1, 63, 1000, 4096])), int(rng.choice([1, 12, 37])), float(rng.choice([0.005, 0.02]))
    g = torch.Generator().manual_seed(int(rng.integers(0, 1 << 30)))
    prev = torch.randn(N, J, generator=g)
    st = ArticulationRootState(N, J, "cuda:0", prev.cuda())
    sim_t, acc_t = 0.0, -1.0
    for k in range(int(rng.integers(2, 6))):
        tf = torch.cat([torch.randn(N, 3, generator=g), torch.nn.functional.normalize(torch.randn(N, 4, generator=g), dim=1)], 1)
        vel, dv = torch.randn(N, 6, generator=g), torch.randn(N, J, generator=g)
        sim_t += dt
        elapsed, acc_t = sim_t - acc_t, sim_t
        st.update(tf.cuda(), vel.cuda(), dv.cuda(), dt)
        assert torch.equal(st.root_pos_w.cpu(), tf[:, :3]) and torch.equal(st.root_quat_w.cpu(), convert_quat(tf[:, 3:7], to="wxyz")), (k, "root pose")
        assert torch.equal(st.root_lin_vel_w.cpu(), vel[:, :3]) and torch.equal(st.root_ang_vel_w.cpu(), vel[:, 3:]), (k, "root velocity")
        assert_close(st.joint_acc, (dv - prev) / elapsed, 1e-5, "joint_acc")
        prev = dv.clone()
    return f"N={N} J={J} dt={dt}"


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for name, fn in (("contact_sensor", case_contact), ("velocity_command", case_command), ("pd_actuator", case_pd_actuator),
                     ("articulation", case_articulation)):
        rng = np.random.default_rng(seed)
        nbad, last = 0, ""
        for c in range(cases):
            try:
                last = fn(rng)
            except (AssertionError, RuntimeError, ValueError) as exc:
                nbad += 1
                print(f"{name} case {c}: FAIL {type(exc).__name__}: {str(exc)[:300]}", flush=True)
        bad += nbad
        print(f"{name}: {cases - nbad} / {cases} cases agree (last: {last})", flush=True)
    sys.exit(1 if bad else 0)
