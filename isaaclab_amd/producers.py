"""Host-side mirrors of the reference's per-step input producers, running on libimx:

* :class:`ContactSensorState` -- ``ContactSensor`` data buffers + ``update(dt)`` (reference
  isaaclab/sensors/contact_sensor/contact_sensor.py:140-210,320-379; isaaclab/sensors/sensor_base.py:182-205,287-297)
* :class:`UniformVelocityCommand` -- ``CommandTerm.reset/compute`` + the uniform velocity command
  (isaaclab/managers/command_manager.py:119-187; isaaclab/envs/mdp/commands/velocity_command.py:37-160)

Attribute names follow the reference so that term functions reading ``sensor.data.*`` / ``command_manager`` keep working.
"""

from __future__ import annotations

import ctypes
import types

import numpy as np
import torch

from . import _lib
from ._lib import check, lib


class ContactSensorState:
    def __init__(self, num_envs: int, num_bodies: int, history_length: int = 0, track_air_time: bool = False,
                 update_period: float = 0.0, force_threshold: float = 1.0, device="cuda:0"):
        N, B, H = num_envs, num_bodies, history_length
        self.num_envs, self.num_bodies, self.history_length = N, B, H
        self.cfg = types.SimpleNamespace(track_air_time=track_air_time, update_period=update_period,
                                         force_threshold=force_threshold, history_length=H)
        dev = torch.device(device)
        self.device = dev
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self._timestamp, self._timestamp_last_update = z(N), z(N)
        self._is_outdated = torch.ones(N, dtype=torch.bool, device=dev)  # sensor_base.py:_initialize_impl
        d = types.SimpleNamespace()
        d.net_forces_w = z(N, B, 3)
        d.net_forces_w_history = z(N, max(H, 1), B, 3)
        d.last_air_time, d.current_air_time = z(N, B), z(N, B)
        d.last_contact_time, d.current_contact_time = z(N, B), z(N, B)
        self.data = d

    def update(self, new_net_forces_w: torch.Tensor, dt: float):
        """``SensorBase.update(dt)`` with PhysX's ``get_net_contact_forces`` result passed in as ``(N,B,3)``."""
        d = self.data
        f = new_net_forces_w.reshape(self.num_envs, self.num_bodies, 3).contiguous()
        H = self.history_length
        check(lib().imx_contact_sensor_update(
            self.num_envs, self.num_bodies, H, f.data_ptr(), float(dt), float(self.cfg.update_period),
            float(self.cfg.force_threshold), int(self.cfg.track_air_time), self._timestamp.data_ptr(),
            self._timestamp_last_update.data_ptr(), self._is_outdated.data_ptr(), d.net_forces_w.data_ptr(),
            d.net_forces_w_history.data_ptr() if H > 0 else None, d.last_air_time.data_ptr(), d.current_air_time.data_ptr(),
            d.last_contact_time.data_ptr(), d.current_contact_time.data_ptr(), _lib.current_stream(self.device)))
        if H == 0:  # contact_sensor.py:302: history is a view of the current forces
            d.net_forces_w_history = d.net_forces_w.unsqueeze(1)

    def reset(self, env_ids=None):
        """contact_sensor.py:143-165 + sensor_base.py:182-194"""
        ids = slice(None) if env_ids is None else env_ids
        self._timestamp[ids] = 0.0
        self._timestamp_last_update[ids] = 0.0
        self._is_outdated[ids] = True
        d = self.data
        d.net_forces_w[ids] = 0.0
        d.net_forces_w_history[ids] = 0.0
        if self.cfg.track_air_time:
            d.current_air_time[ids] = 0.0
            d.last_air_time[ids] = 0.0
            d.current_contact_time[ids] = 0.0
            d.last_contact_time[ids] = 0.0

    def compute_first_contact(self, dt: float, abs_tol: float = 1.0e-8) -> torch.Tensor:
        """contact_sensor.py:176-210"""
        if not self.cfg.track_air_time:
            raise RuntimeError("The contact sensor is not configured to track contact time."
                               "Please enable the 'track_air_time' in the sensor configuration.")
        c = self.data.current_contact_time
        return (c > 0.0) * (c < (dt + abs_tol))


class UniformVelocityCommand:
    """``cfg``: dict / object with the fields of ``UniformVelocityCommandCfg`` (velocity_command_cfg.py)."""

    def __init__(self, cfg, num_envs: int, step_dt: float, device="cuda:0", seed: int = 0):
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        rng = get("ranges")
        rget = (lambda k: rng.get(k)) if isinstance(rng, dict) else (lambda k: getattr(rng, k))
        self.heading_command = bool(get("heading_command", False))
        heading = rget("heading")
        if self.heading_command and heading is None:
            raise ValueError("The velocity command has heading commands active (heading_command=True) but the "
                             "`ranges.heading` parameter is set to None.")
        heading = heading or (0.0, 0.0)
        rt = get("resampling_time_range")
        self._cfg15 = np.asarray([rt[0], rt[1], *rget("lin_vel_x"), *rget("lin_vel_y"), *rget("ang_vel_z"), *heading,
                                  get("rel_standing_envs", 0.0), get("rel_heading_envs", 1.0),
                                  get("heading_control_stiffness", 1.0), rt[1] / step_dt, 0.0], dtype=np.float32)
        N = num_envs
        dev = torch.device(device)
        self.num_envs, self.device, self.seed = N, dev, int(seed)
        self.vel_command_b = torch.zeros(N, 3, device=dev)
        self.heading_target = torch.zeros(N, device=dev)
        self.is_heading_env = torch.zeros(N, dtype=torch.bool, device=dev)
        self.is_standing_env = torch.zeros(N, dtype=torch.bool, device=dev)
        self.time_left = torch.zeros(N, device=dev)
        self.command_counter = torch.zeros(N, dtype=torch.long, device=dev)
        self.metrics = {"error_vel_xy": torch.zeros(N, device=dev), "error_vel_yaw": torch.zeros(N, device=dev)}
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)

    @property
    def command(self) -> torch.Tensor:
        return self.vel_command_b

    def compute(self, dt: float, root_quat_w, root_lin_vel_w, root_ang_vel_w, reset_mask=None, uniforms=None,
                do_compute: bool = True):
        """``reset(ids of reset_mask)`` then (``do_compute``) ``compute(dt)``; ``uniforms``: optional (2,N,7) parity samples."""
        self._step += 1
        check(lib().imx_velocity_command(
            self.num_envs, self._cfg15.ctypes.data, int(self.heading_command), float(dt), int(do_compute), root_quat_w.data_ptr(),
            root_lin_vel_w.data_ptr(), root_ang_vel_w.data_ptr(), _lib.ptr(reset_mask), _lib.ptr(uniforms), self.seed,
            self._step.data_ptr(), self.vel_command_b.data_ptr(), self.heading_target.data_ptr(),
            self.is_heading_env.data_ptr(), self.is_standing_env.data_ptr(), self.time_left.data_ptr(),
            self.command_counter.data_ptr(), self.metrics["error_vel_xy"].data_ptr(),
            self.metrics["error_vel_yaw"].data_ptr(), _lib.current_stream(self.device)))
        return self.vel_command_b


class ArticulationRootState:
    """``ArticulationData`` root state + ``joint_acc`` from PhysX-layout tensors (reference
    isaaclab/assets/articulation/articulation_data.py:365-380,546-556), SURVEY 8f row 4.

    ``update(root_transforms (N,7 pos+quat XYZW), root_velocities (N,6), dof_velocities (N,J), dt)`` fills
    ``root_pos_w, root_quat_w (WXYZ), root_lin_vel_w, root_ang_vel_w, joint_vel, joint_acc``.
    """

    def __init__(self, num_envs: int, num_joints: int, device="cuda:0", initial_joint_vel: torch.Tensor | None = None):
        N, J = num_envs, num_joints
        dev = torch.device(device)
        self.num_envs, self.num_joints, self.device = N, J, dev
        self.root_pos_w = torch.zeros(N, 3, device=dev)
        self.root_quat_w = torch.zeros(N, 4, device=dev)
        self.root_lin_vel_w = torch.zeros(N, 3, device=dev)
        self.root_ang_vel_w = torch.zeros(N, 3, device=dev)
        self.joint_acc = torch.zeros(N, J, device=dev)
        self._previous_joint_vel = torch.zeros(N, J, device=dev) if initial_joint_vel is None else initial_joint_vel.clone()
        self.joint_vel = self._previous_joint_vel
        # TimestampedBuffer semantics (isaaclab/utils/buffers/timestamped_buffer.py): buffers start at timestamp -1.0, so
        # the reference's first finite difference divides by (dt + 1.0) -- reproduced, not "fixed"
        self._sim_timestamp = 0.0
        self._joint_acc_timestamp = -1.0

    def update(self, root_transforms, root_velocities, dof_velocities, dt: float):
        self._sim_timestamp += dt
        elapsed = self._sim_timestamp - self._joint_acc_timestamp
        self._joint_acc_timestamp = self._sim_timestamp
        check(lib().imx_articulation_update(
            self.num_envs, self.num_joints, root_transforms.data_ptr(), root_velocities.data_ptr(), dof_velocities.data_ptr(),
            float(elapsed), self.root_pos_w.data_ptr(), self.root_quat_w.data_ptr(), self.root_lin_vel_w.data_ptr(),
            self.root_ang_vel_w.data_ptr(), self._previous_joint_vel.data_ptr(), self.joint_acc.data_ptr(),
            _lib.current_stream(self.device)))
        self.joint_vel = dof_velocities


class PDActuator:
    """``IdealPDActuator`` / ``ImplicitActuator`` (reporting) / ``DCMotor`` ``.compute`` on libimx
    (reference isaaclab/actuators/actuator_pd.py:115-145,184-199,264-286): ``computed_effort`` and ``applied_effort`` (N,J)."""

    def __init__(self, stiffness, damping, effort_limit, velocity_limit=None, saturation_effort: float | None = None):
        self.stiffness, self.damping, self.effort_limit = stiffness.contiguous(), damping.contiguous(), effort_limit.contiguous()
        self.velocity_limit = None if velocity_limit is None else velocity_limit.contiguous()
        self.saturation_effort = saturation_effort
        if stiffness.device.type != "cuda":
            raise RuntimeError("PDActuator needs a GPU: libimx has no CPU path")
        self.computed_effort = torch.zeros_like(self.stiffness)
        self.applied_effort = torch.zeros_like(self.stiffness)

    def compute(self, joint_pos_target, joint_pos, joint_vel, joint_vel_target=None, effort_ff=None):
        N, J = self.stiffness.shape
        p = _lib.ptr
        dc = self.saturation_effort is not None
        check(lib().imx_actuator_pd(N, J, int(dc), float(self.saturation_effort or 0.0), p(joint_pos_target), p(joint_vel_target),
                                    p(effort_ff), p(joint_pos), p(joint_vel), p(self.stiffness), p(self.damping), p(self.effort_limit),
                                    p(self.velocity_limit), p(self.computed_effort), p(self.applied_effort),
                                    _lib.current_stream(self.stiffness.device)))
        return self.applied_effort


class DelayedPDActuator:
    """``DelayedPDActuator`` / ``RemotizedPDActuator`` ``.reset`` + ``.compute`` on libimx (reference
    isaaclab/actuators/actuator_pd.py:289-412): the position / velocity / feed-forward commands pass through one shared delay ring
    with a per-env lag re-drawn at reset, then the PD law; ``joint_parameter_lookup`` (K,3) adds the remotized angle-dependent
    torque limit.  One launch per physics step, no host sync (the reference's ``CircularBuffer.append`` has a ``.tolist()``)."""

    def __init__(self, stiffness, damping, min_delay: int, max_delay: int, effort_limit=None, joint_parameter_lookup=None):
        if stiffness.device.type != "cuda":
            raise RuntimeError("DelayedPDActuator needs a GPU: libimx has no CPU path")
        if not 0 <= int(min_delay) <= int(max_delay):
            raise ValueError(f"The minimum time lag cannot be negative or above the maximum. Received: {min_delay}, {max_delay}")
        dev = stiffness.device
        N, J = stiffness.shape
        self.stiffness, self.damping = stiffness.contiguous(), damping.contiguous()
        self.min_delay, self.max_delay = int(min_delay), int(max_delay)
        self.lookup = None
        if joint_parameter_lookup is not None:  # RemotizedPDActuator removes the box limits (:373-380)
            self.lookup = torch.as_tensor(joint_parameter_lookup, dtype=torch.float32, device=dev).contiguous()
            x = self.lookup[:, 0]
            if self.lookup.numel() == 0:
                raise ValueError("Input tensor x is empty!")
            if bool(torch.any(x[1:] < x[:-1])):
                raise ValueError("Input tensor x is not sorted in ascending order!")
            effort_limit = None
        self.effort_limit = None if effort_limit is None else effort_limit.contiguous()
        self.ring = torch.zeros(self.max_delay + 1, N, 3, J, device=dev)
        self.time_lags = torch.zeros(N, dtype=torch.int32, device=dev)
        self.reset_step = torch.zeros(N, dtype=torch.int64, device=dev)
        self.step = 0
        self.computed_effort = torch.zeros_like(self.stiffness)
        self.applied_effort = torch.zeros_like(self.stiffness)

    def reset(self, env_ids=None, time_lags=None):
        """New random lag in [min_delay, max_delay] for ``env_ids`` (``time_lags`` overrides the draw: parity tests) and an empty
        delay line: the next sample fills it."""
        N = self.stiffness.shape[0]
        ids = slice(None) if env_ids is None else env_ids
        n = N if env_ids is None else len(env_ids)
        if time_lags is None:
            time_lags = torch.randint(self.min_delay, self.max_delay + 1, (n,), dtype=torch.int32, device=self.stiffness.device)
        self.time_lags[ids] = time_lags.to(self.time_lags)
        self.reset_step[ids] = self.step

    def compute(self, joint_pos_target, joint_pos, joint_vel, joint_vel_target=None, effort_ff=None):
        N, J = self.stiffness.shape
        p = _lib.ptr
        check(lib().imx_actuator_delayed_pd(N, J, self.max_delay, self.step, p(self.time_lags), p(self.reset_step), p(self.ring),
                                            p(joint_pos_target), p(joint_vel_target), p(effort_ff), p(joint_pos), p(joint_vel),
                                            p(self.stiffness), p(self.damping), p(self.effort_limit), p(self.lookup),
                                            0 if self.lookup is None else self.lookup.shape[0], p(self.computed_effort),
                                            p(self.applied_effort), _lib.current_stream(self.stiffness.device)))
        self.step += 1
        return self.applied_effort
