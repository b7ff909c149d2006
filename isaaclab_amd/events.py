"""Reset / interval events and the terrain curriculum on libimx.

Host-side mirror of the reference's ``EventManager.apply(mode="reset" | "interval")`` for the locomotion tasks' terms
(reference ``isaaclab/envs/mdp/events.py``: ``reset_root_state_uniform`` :823, ``reset_joints_by_scale`` :987,
``reset_joints_by_offset`` :1020, ``push_by_setting_velocity`` :795) and of ``terrain_levels_vel``
(``isaaclab_tasks/.../locomotion/velocity/mdp/curriculums.py:26-55``).  The reference calls each term with a compacted
``env_ids`` tensor; here every call takes the boolean reset mask the step kernel already produced and rewrites only the
flagged rows of the caller's "to simulator" buffers -- no ``nonzero``, no ``len(env_ids)`` host sync.  No CPU fallback.
"""

from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check, lib

AXES = ("x", "y", "z", "roll", "pitch", "yaw")


def _axis_ranges(d) -> list[float]:
    d = d or {}
    out = []
    for k in AXES:
        lo, hi = d.get(k, (0.0, 0.0))
        out += [float(lo), float(hi)]
    return out


def _func_name(term: dict) -> str:
    f = term.get("func", "")
    f = f if isinstance(f, str) else getattr(f, "__name__", "")
    return f.replace(":", ".").rsplit(".", 1)[-1]


class ResetEvents:
    """``reset_root_state_uniform`` + ``reset_joints_by_scale|offset`` (one launch) and ``push_by_setting_velocity``."""

    def __init__(self, num_envs: int, num_joints: int, device, pose_range=None, velocity_range=None,
                 joint_position_range=(1.0, 1.0), joint_velocity_range=(0.0, 0.0), joint_mode: str | None = "scale",
                 push_velocity_range=None, seed: int = 0):
        self.N, self.J, self.device = int(num_envs), int(num_joints), torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ResetEvents needs a GPU: libimx has no CPU path")
        self.joint_mode = {"scale": 0, "offset": 1, None: -1}[joint_mode]
        r = _axis_ranges(pose_range) + _axis_ranges(velocity_range) + [float(x) for x in (*joint_position_range, *joint_velocity_range)]
        self._ranges28 = (ctypes.c_float * 28)(*r)
        self._push12 = (ctypes.c_float * 12)(*_axis_ranges(push_velocity_range)) if push_velocity_range is not None else None
        self.seed = int(seed)
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)  # advances the counter-based generator

    @classmethod
    def from_cfg(cls, events_cfg: dict, num_envs: int, num_joints: int, device, seed: int = 0) -> "ResetEvents":
        """``events_cfg``: the reference's ``EventCfg`` in ``to_dict()`` form ({term: {func, mode, params}})."""
        kw = dict(joint_mode=None)
        for name, term in (events_cfg or {}).items():
            if not isinstance(term, dict):
                continue
            fn, params = _func_name(term), term.get("params", {}) or {}
            if fn == "reset_root_state_uniform":
                kw.update(pose_range=params.get("pose_range"), velocity_range=params.get("velocity_range"))
            elif fn in ("reset_joints_by_scale", "reset_joints_by_offset"):
                kw.update(joint_position_range=tuple(params["position_range"]), joint_velocity_range=tuple(params["velocity_range"]),
                          joint_mode="scale" if fn.endswith("scale") else "offset")
            elif fn == "push_by_setting_velocity":
                kw.update(push_velocity_range=params.get("velocity_range"))
        return cls(num_envs, num_joints, device, seed=seed, **kw)

    def reset(self, reset_mask, default_root_state, env_origins, root_pose, root_vel, default_joint_pos=None, default_joint_vel=None,
              soft_joint_pos_limits=None, soft_joint_vel_limits=None, joint_pos=None, joint_vel=None, uniforms=None):
        """Rewrites rows ``reset_mask != 0`` of root_pose (N,7), root_vel (N,6), joint_pos / joint_vel (N,J) in place."""
        p = _lib.ptr
        check(lib().imx_reset_events(self.N, self.J, p(reset_mask), self._ranges28, self.joint_mode, p(default_root_state), p(env_origins),
                                     p(default_joint_pos), p(default_joint_vel), p(soft_joint_pos_limits), p(soft_joint_vel_limits),
                                     p(uniforms), self.seed, self._step.data_ptr(), p(root_pose), p(root_vel), p(joint_pos), p(joint_vel),
                                     _lib.current_stream(self.device)))
        if uniforms is None:
            self._step += 1

    def push(self, mask, root_vel_w, uniforms=None):
        """``push_by_setting_velocity``: root_vel_w (N,6) += U(range) on rows ``mask != 0`` (None = all)."""
        if self._push12 is None:
            raise RuntimeError("no push_by_setting_velocity term configured")
        p = _lib.ptr
        check(lib().imx_push_velocity(self.N, p(mask), self._push12, p(uniforms), self.seed, self._step.data_ptr(), p(root_vel_w),
                                      _lib.current_stream(self.device)))
        if uniforms is None:
            self._step += 1


class ExternalForceTorque:
    """``apply_external_force_torque`` (events.py:764-791) on a subset of bodies, masked."""

    def __init__(self, num_envs: int, num_bodies: int, device, force_range, torque_range, body_ids=None, seed: int = 0):
        self.N, self.NB, self.device = int(num_envs), int(num_bodies), torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ExternalForceTorque needs a GPU: libimx has no CPU path")
        self._ranges = (ctypes.c_float * 4)(float(force_range[0]), float(force_range[1]), float(torque_range[0]), float(torque_range[1]))
        self.body_ids = None if body_ids is None else torch.as_tensor(list(body_ids), dtype=torch.int32, device=self.device)
        self.seed = int(seed)
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)

    def apply(self, mask, forces, torques, uniforms=None):
        """Rewrites rows ``mask != 0`` (None = all) of forces / torques (N, num_bodies, 3) for the selected bodies."""
        p = _lib.ptr
        n = 0 if self.body_ids is None else int(self.body_ids.numel())
        check(lib().imx_external_force_torque(self.N, self.NB, p(mask), p(self.body_ids), n, self._ranges, p(uniforms), self.seed,
                                              self._step.data_ptr(), p(forces), p(torques), _lib.current_stream(self.device)))
        if uniforms is None:
            self._step += 1


class TerrainCurriculum:
    """``terrain_levels_vel`` + ``TerrainImporter.update_env_origins`` (levels, origins updated in place)."""

    def __init__(self, terrain_origins: torch.Tensor, terrain_levels: torch.Tensor, terrain_types: torch.Tensor, env_origins: torch.Tensor,
                 terrain_size_x: float, max_episode_length_s: float, seed: int = 0):
        self.terrain_origins = terrain_origins.contiguous()
        self.terrain_levels, self.terrain_types, self.env_origins = terrain_levels, terrain_types, env_origins
        self.size_x, self.max_len_s, self.seed = float(terrain_size_x), float(max_episode_length_s), int(seed)
        self.device = terrain_origins.device
        if self.device.type != "cuda":
            raise RuntimeError("TerrainCurriculum needs a GPU: libimx has no CPU path")
        self.mean_level = torch.zeros(1, device=self.device)
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)

    def update(self, reset_mask, root_pos_w, command, rand_levels=None) -> torch.Tensor:
        R, C = self.terrain_origins.shape[:2]
        p = _lib.ptr
        check(lib().imx_terrain_levels(self.terrain_levels.shape[0], R, C, p(reset_mask), p(root_pos_w), p(command), p(self.terrain_origins),
                                       p(self.terrain_types), self.size_x, self.max_len_s, p(rand_levels), self.seed,
                                       self._step.data_ptr(), p(self.terrain_levels), p(self.env_origins), p(self.mean_level),
                                       _lib.current_stream(self.device)))
        if rand_levels is None:
            self._step += 1
        return self.mean_level  # 0-dim-like device tensor (the reference returns torch.mean(...) and .item()s it later)
