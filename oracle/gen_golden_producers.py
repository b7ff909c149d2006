"""TEST INFRASTRUCTURE (build container only): tests/golden/producers.npz from the REAL reference classes
``ContactSensor`` (SensorBase.update -> _update_buffers_impl) and ``UniformVelocityCommand`` (CommandTerm.reset/compute),
objects created with ``__new__`` and their buffers set by hand (no PhysX); see oracle/gen_golden.py for the import stub.

Tensor.uniform_ draws of the command term are replaced by a recorded table U[draw, env, column] (column = order of the
uniform_ calls inside CommandTerm._resample / _resample_command) so that the HIP path can be fed the same samples.
"""

from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

ref_import.install()

from isaaclab.assets.articulation.articulation_data import ArticulationData  # noqa: E402
from isaaclab.envs.mdp.commands.velocity_command import UniformVelocityCommand  # noqa: E402
from isaaclab.sensors.contact_sensor import ContactSensor  # noqa: E402
from isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.rough_env_cfg import AnymalCRoughEnvCfg  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def contact_sensor_golden(rec):
    N, B, H, steps = 48, 17, 3, 12
    g = torch.Generator().manual_seed(11)
    s = ContactSensor.__new__(ContactSensor)
    s.cfg = types.SimpleNamespace(history_length=H, track_air_time=True, update_period=0.005, force_threshold=1.0,
                                  filter_prim_paths_expr=[], track_pose=False)
    s._num_envs, s._num_bodies, s._device = N, B, "cpu"
    s._sim_physics_dt = 0.005
    s._is_visualizing = False
    s._timestamp = torch.zeros(N)
    s._timestamp_last_update = torch.zeros(N)
    s._is_outdated = torch.ones(N, dtype=torch.bool)
    s._data = types.SimpleNamespace(net_forces_w=torch.zeros(N, B, 3), net_forces_w_history=torch.zeros(N, H, B, 3),
                                    last_air_time=torch.zeros(N, B), current_air_time=torch.zeros(N, B),
                                    last_contact_time=torch.zeros(N, B), current_contact_time=torch.zeros(N, B))
    forces = []
    holder = {}
    s._contact_physx_view = types.SimpleNamespace(get_net_contact_forces=lambda dt: holder["f"], filter_count=0)
    # contact pattern: bodies toggle between contact / air with random dwell times; magnitudes around the threshold too
    state = torch.rand(N, B, generator=g) < 0.5
    for t in range(steps):
        flip = torch.rand(N, B, generator=g) < 0.25
        state = state ^ flip
        mag = torch.where(state, torch.rand(N, B, generator=g) * 50 + 0.5, torch.rand(N, B, generator=g) * 0.9)
        d = torch.randn(N, B, 3, generator=g)
        f = d / d.norm(dim=-1, keepdim=True) * mag.unsqueeze(-1)
        forces.append(f.clone())
        holder["f"] = f.view(-1, 3)
        if t == 7:  # reset a few envs mid-way (contact_sensor.py:143-165)
            ContactSensor.reset(s, torch.tensor([1, 5, 30]))
        s.update(0.005)
        for k in ("net_forces_w", "net_forces_w_history", "last_air_time", "current_air_time", "last_contact_time",
                  "current_contact_time"):
            rec[f"contact/step{t}/{k}"] = getattr(s._data, k).numpy().copy()
        rec[f"contact/step{t}/timestamp"] = s._timestamp.numpy().copy()
        rec[f"contact/step{t}/timestamp_last_update"] = s._timestamp_last_update.numpy().copy()
        rec[f"contact/step{t}/first_contact"] = s.compute_first_contact(0.02).numpy().copy()
    rec["contact/forces"] = torch.stack(forces).numpy()
    rec["contact/meta"] = np.array(json.dumps(dict(N=N, B=B, H=H, steps=steps, dt=0.005, update_period=0.005,
                                                   force_threshold=1.0, reset_step=7, reset_ids=[1, 5, 30])))


class FakeRobotData:
    root_lin_vel_b = ArticulationData.root_lin_vel_b
    root_ang_vel_b = ArticulationData.root_ang_vel_b
    heading_w = ArticulationData.heading_w

    def __init__(self, N):
        self.FORWARD_VEC_B = torch.tensor((1.0, 0.0, 0.0)).repeat(N, 1)


def command_golden(rec):
    N, steps, step_dt = 64, 8, 0.02
    cfg = AnymalCRoughEnvCfg().commands.base_velocity
    cfg.resampling_time_range = (0.05, 0.09)  # short, so that timer-driven resampling happens within the fixture
    cfg.rel_standing_envs = 0.2
    cfg.rel_heading_envs = 0.7
    g = torch.Generator().manual_seed(13)
    term = UniformVelocityCommand.__new__(UniformVelocityCommand)
    term.cfg = cfg
    term._debug_vis_handle = None
    term._env = types.SimpleNamespace(num_envs=N, device="cpu", step_dt=step_dt)
    data = FakeRobotData(N)
    term.robot = types.SimpleNamespace(data=data)
    term.metrics = {"error_vel_xy": torch.zeros(N), "error_vel_yaw": torch.zeros(N)}
    term.time_left = torch.zeros(N)
    term.command_counter = torch.zeros(N, dtype=torch.long)
    term.vel_command_b = torch.zeros(N, 3)
    term.heading_target = torch.zeros(N)
    term.is_heading_env = torch.zeros(N, dtype=torch.bool)
    term.is_standing_env = torch.zeros(N, dtype=torch.bool)

    ctx = {"ids": None, "col": 0, "U": None, "draw": torch.zeros(N, dtype=torch.long)}
    real_uniform = torch.Tensor.uniform_
    real_resample = UniformVelocityCommand._resample

    def fake_uniform(self, lo=0.0, hi=1.0):
        ids, col = ctx["ids"], ctx["col"]
        ctx["col"] += 1
        u = ctx["U"][ctx["draw"][ids], ids, col]
        self.copy_(u * (hi - lo) + lo)
        return self

    def wrapped_resample(self, env_ids):
        env_ids = torch.as_tensor(env_ids)
        if len(env_ids) == 0:
            return
        ctx["ids"], ctx["col"] = env_ids, 0
        real_resample(self, env_ids)
        ctx["draw"][env_ids] += 1

    torch.Tensor.uniform_ = fake_uniform
    UniformVelocityCommand._resample = wrapped_resample
    try:
        for t in range(steps):
            q = torch.randn(N, 4, generator=g)
            q = q / q.norm(dim=-1, keepdim=True)
            data.root_quat_w = q
            data.root_link_quat_w = q
            data.root_lin_vel_w = torch.randn(N, 3, generator=g) * 0.5
            data.root_ang_vel_w = torch.randn(N, 3, generator=g) * 0.5
            U = torch.rand(2, N, 7, generator=g)
            reset_mask = torch.rand(N, generator=g) < (1.0 if t == 0 else 0.1)
            ctx["U"] = U
            ctx["draw"][:] = 0
            ids = reset_mask.nonzero().flatten()
            if len(ids):
                term.reset(ids)
            term.compute(step_dt)
            tag = f"command/step{t}"
            rec[f"{tag}/root_quat_w"], rec[f"{tag}/root_lin_vel_w"], rec[f"{tag}/root_ang_vel_w"] = (
                q.numpy().copy(), data.root_lin_vel_w.numpy().copy(), data.root_ang_vel_w.numpy().copy())
            rec[f"{tag}/uniforms"] = U.numpy().copy()
            rec[f"{tag}/reset_mask"] = reset_mask.numpy().copy()
            for k in ("vel_command_b", "heading_target", "is_heading_env", "is_standing_env", "time_left", "command_counter"):
                rec[f"{tag}/{k}"] = getattr(term, k).numpy().copy()
            rec[f"{tag}/error_vel_xy"] = term.metrics["error_vel_xy"].numpy().copy()
            rec[f"{tag}/error_vel_yaw"] = term.metrics["error_vel_yaw"].numpy().copy()
    finally:
        torch.Tensor.uniform_ = real_uniform
        UniformVelocityCommand._resample = real_resample
    d = cfg.to_dict()
    keep = {k: d[k] for k in ("resampling_time_range", "heading_command", "heading_control_stiffness", "rel_standing_envs",
                              "rel_heading_envs", "ranges")}
    rec["command/meta"] = np.array(json.dumps(dict(N=N, steps=steps, step_dt=step_dt, cfg=keep), default=list))


def articulation_golden(rec):
    """Real ArticulationData over a fake PhysX view (root transforms in XYZW, velocities, dof velocities)."""
    import isaaclab.assets.articulation.articulation_data as ad_mod

    N, J, steps, dt = 32, 12, 4, 0.005
    g = torch.Generator().manual_seed(17)
    cur = {}
    class View:  # weak-referenceable stand-in for physx.ArticulationView
        count = N
        get_root_transforms = staticmethod(lambda: cur["tf"])
        get_root_velocities = staticmethod(lambda: cur["vel"])
        get_dof_velocities = staticmethod(lambda: cur["dv"])
        get_dof_positions = staticmethod(lambda: cur["dv"])

    view = View()
    ad_mod.SimulationManager = types.SimpleNamespace(
        get_physics_sim_view=lambda: types.SimpleNamespace(get_gravity=lambda: (0.0, 0.0, -9.81)))
    cur["dv"] = torch.randn(N, J, generator=g)
    cur["tf"] = torch.zeros(N, 7)
    cur["vel"] = torch.zeros(N, 6)
    rec["artic/initial_joint_vel"] = cur["dv"].numpy().copy()
    data = ArticulationData(view, "cpu")
    for t in range(steps):
        q = torch.randn(N, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        cur["tf"] = torch.cat([torch.randn(N, 3, generator=g), q], dim=-1)  # PhysX: quat is XYZW
        cur["vel"] = torch.randn(N, 6, generator=g)
        cur["dv"] = torch.randn(N, J, generator=g)
        data.update(dt)
        tag = f"artic/step{t}"
        rec[f"{tag}/root_transforms"], rec[f"{tag}/root_velocities"], rec[f"{tag}/dof_velocities"] = (
            cur["tf"].numpy().copy(), cur["vel"].numpy().copy(), cur["dv"].numpy().copy())
        rec[f"{tag}/root_pos_w"] = data.root_pos_w.numpy().copy()
        rec[f"{tag}/root_quat_w"] = data.root_quat_w.numpy().copy()
        rec[f"{tag}/root_lin_vel_w"] = data.root_lin_vel_w.numpy().copy()
        rec[f"{tag}/root_ang_vel_w"] = data.root_ang_vel_w.numpy().copy()
        rec[f"{tag}/joint_acc"] = data.joint_acc.numpy().copy()
    rec["artic/meta"] = np.array(json.dumps(dict(N=N, J=J, steps=steps, dt=dt)))


def main():
    rec = {}
    contact_sensor_golden(rec)
    command_golden(rec)
    articulation_golden(rec)
    np.savez_compressed(os.path.join(GOLDEN, "producers.npz"), **rec)
    print("[golden] producers:", len(rec), "arrays")


if __name__ == "__main__":
    main()
