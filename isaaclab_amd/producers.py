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


_NET_ACT = {"identity": 0, None: 0, "softsign": 1, "tanh": 2, "relu": 3, "elu": 4}


def _pack_dense(layers) -> tuple[list[torch.Tensor], list[int]]:
    """[(weight (out,in), bias (out)), ...] -> flat pieces + widths."""
    pieces, widths = [], []
    for w, b in layers:
        pieces += [w.detach().float().reshape(-1), b.detach().float().reshape(-1)]
        widths.append(int(w.shape[0]))
    return pieces, widths


def _dense_from_module(net, skip_prefix: str | None) -> list:
    """The Linear layers of a (TorchScript) network in definition order: every 2-D ``*.weight`` with its ``*.bias``."""
    sd = net.state_dict()
    layers = []
    for name, w in sd.items():
        if name.endswith("weight") and w.dim() == 2 and not (skip_prefix and name.startswith(skip_prefix)):
            layers.append((w, sd[name[: -len("weight")] + "bias"]))
    if not layers:
        raise ValueError("no Linear layers found in the actuator network")
    return layers


class ActuatorNetLSTM:
    """``ActuatorNetLSTM.reset`` / ``.compute`` on libimx (reference isaaclab/actuators/actuator_net.py:29-104; the ANYdrive 3 actuator
    of ANYmal-B/C, isaaclab_assets/robots/anymal.py:45-51): every (env, joint) sample runs through the LSTM stack and the dense head
    in one launch; hidden and cell states live in ``sea_hidden_state`` / ``sea_cell_state`` (num_layers, N*J, H) like the reference's.

    ``network``: the reference's TorchScript module (``network.lstm`` = ``nn.LSTM(2, H, L, batch_first=True)``, the remaining Linear
    layers = the head, ``head_activation`` between them) or ``None`` with ``lstm_layers`` = [(w_ih, w_hh, b_ih, b_hh), ...] and
    ``head`` = [(weight, bias), ...] given directly."""

    def __init__(self, num_envs: int, num_joints: int, effort_limit, velocity_limit, saturation_effort: float, network=None,
                 lstm_layers=None, head=None, head_activation: str | None = "softsign", device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ActuatorNetLSTM needs a GPU: libimx has no CPU path")
        if network is not None:
            sd = network.lstm.state_dict()
            L = len(sd) // 4  # actuator_net.py:54
            lstm_layers = [(sd[f"weight_ih_l{k}"], sd[f"weight_hh_l{k}"], sd[f"bias_ih_l{k}"], sd[f"bias_hh_l{k}"]) for k in range(L)]
            head = _dense_from_module(network, "lstm.")
        if not lstm_layers or not head:
            raise ValueError("ActuatorNetLSTM needs a network (or lstm_layers and head)")
        self.num_envs, self.num_joints = int(num_envs), int(num_joints)
        self.hidden_dim = int(lstm_layers[0][1].shape[1])
        self.num_layers = len(lstm_layers)
        pieces = []
        for w_ih, w_hh, b_ih, b_hh in lstm_layers:
            pieces += [t.detach().float().reshape(-1) for t in (w_ih, w_hh, b_ih, b_hh)]
        hp, self._dense_out = _pack_dense(head)
        self._weights = torch.cat(pieces + hp).to(dev).contiguous()
        self._dense_arr = (ctypes.c_int32 * len(self._dense_out))(*self._dense_out)
        self._act = _NET_ACT[head_activation]
        full = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev).expand(self.num_envs, self.num_joints).contiguous()  # noqa: E731
        self.effort_limit, self.velocity_limit = full(effort_limit), full(velocity_limit)
        self.saturation_effort = float(saturation_effort)
        n = self.num_envs * self.num_joints
        self.sea_hidden_state = torch.zeros(self.num_layers, n, self.hidden_dim, device=dev)
        self.sea_cell_state = torch.zeros(self.num_layers, n, self.hidden_dim, device=dev)
        shape = (self.num_layers, self.num_envs, self.num_joints, self.hidden_dim)
        self.sea_hidden_state_per_env = self.sea_hidden_state.view(shape)
        self.sea_cell_state_per_env = self.sea_cell_state.view(shape)
        self.computed_effort = torch.zeros(self.num_envs, self.num_joints, device=dev)
        self.applied_effort = torch.zeros(self.num_envs, self.num_joints, device=dev)

    def reset(self, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        self.sea_hidden_state_per_env[:, ids] = 0.0
        self.sea_cell_state_per_env[:, ids] = 0.0

    def compute(self, joint_pos_target, joint_pos, joint_vel):
        p = _lib.ptr
        check(lib().imx_actuator_net_lstm(self.num_envs, self.num_joints, self.num_layers, self.hidden_dim, len(self._dense_out),
                                          self._dense_arr, self._act, p(self._weights), self._weights.numel(), p(joint_pos_target),
                                          p(joint_pos), p(joint_vel), p(self.sea_hidden_state), p(self.sea_cell_state),
                                          self.saturation_effort, p(self.effort_limit), p(self.velocity_limit), p(self.computed_effort),
                                          p(self.applied_effort), _lib.current_stream(self._weights.device)))
        return self.applied_effort


class ActuatorNetMLP:
    """``ActuatorNetMLP.reset`` / ``.compute`` on libimx (reference isaaclab/actuators/actuator_net.py:107-195): the position-error and
    velocity histories (N, history, J) are rolled and topped up, the ``input_idx`` entries of both scaled and stacked per
    (env, joint) sample, the dense network evaluated and its torque clipped by the DC-motor model -- one launch.
    ``network``: the TorchScript module (its Linear layers in order, ``activation`` between them) or ``None`` with ``layers`` given."""

    def __init__(self, num_envs: int, num_joints: int, effort_limit, velocity_limit, saturation_effort: float, input_idx, pos_scale: float,
                 vel_scale: float, torque_scale: float, input_order: str = "pos_vel", network=None, layers=None,
                 activation: str | None = "softsign", device="cuda:0"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("ActuatorNetMLP needs a GPU: libimx has no CPU path")
        if input_order not in ("pos_vel", "vel_pos"):  # actuator_net.py:186-189
            raise ValueError(f"Invalid input order for MLP actuator net: {input_order}. Must be 'pos_vel' or 'vel_pos'.")
        if network is not None:
            layers = _dense_from_module(network, None)
        if not layers:
            raise ValueError("ActuatorNetMLP needs a network (or layers)")
        self.num_envs, self.num_joints = int(num_envs), int(num_joints)
        self.input_idx = [int(i) for i in input_idx]
        self.history_length = max(self.input_idx) + 1  # :145
        pieces, self._dense_out = _pack_dense(layers)
        self._weights = torch.cat(pieces).to(dev).contiguous()
        self._dense_arr = (ctypes.c_int32 * len(self._dense_out))(*self._dense_out)
        self._act = _NET_ACT[activation]
        self._idx = torch.tensor(self.input_idx, dtype=torch.int32, device=dev)
        self.pos_scale, self.vel_scale, self.torque_scale = float(pos_scale), float(vel_scale), float(torque_scale)
        self._vel_first = int(input_order == "vel_pos")
        full = lambda v: torch.as_tensor(v, dtype=torch.float32, device=dev).expand(self.num_envs, self.num_joints).contiguous()  # noqa: E731
        self.effort_limit, self.velocity_limit = full(effort_limit), full(velocity_limit)
        self.saturation_effort = float(saturation_effort)
        self._joint_pos_error_history = torch.zeros(self.num_envs, self.history_length, self.num_joints, device=dev)
        self._joint_vel_history = torch.zeros(self.num_envs, self.history_length, self.num_joints, device=dev)
        self.computed_effort = torch.zeros(self.num_envs, self.num_joints, device=dev)
        self.applied_effort = torch.zeros(self.num_envs, self.num_joints, device=dev)

    def reset(self, env_ids=None):
        ids = slice(None) if env_ids is None else env_ids
        self._joint_pos_error_history[ids] = 0.0
        self._joint_vel_history[ids] = 0.0

    def compute(self, joint_pos_target, joint_pos, joint_vel):
        p = _lib.ptr
        check(lib().imx_actuator_net_mlp(self.num_envs, self.num_joints, len(self._dense_out), self._dense_arr, self._act, p(self._weights),
                                         self._weights.numel(), self.history_length, p(self._idx), len(self.input_idx), self.pos_scale,
                                         self.vel_scale, self.torque_scale, self._vel_first, p(joint_pos_target), p(joint_pos), p(joint_vel),
                                         p(self._joint_pos_error_history), p(self._joint_vel_history), self.saturation_effort,
                                         p(self.effort_limit), p(self.velocity_limit), p(self.computed_effort), p(self.applied_effort),
                                         _lib.current_stream(self._weights.device)))
        return self.applied_effort
