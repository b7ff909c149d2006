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


# ---------------------------------------------------------------------------------------------------- orchestration (imx_reset_orchestrate)
_EVENT_OPS = {"reset_root_state_uniform": 1, "reset_joints_by_scale": 2, "reset_joints_by_offset": 3, "push_by_setting_velocity": 4,
              "apply_external_force_torque": 5}
_INTERVAL_OK = ("push_by_setting_velocity", "apply_external_force_torque")


class EventTermState:
    """One reset- or interval-mode term of the reference's ``EventCfg``: its compiled ranges and the state ``EventManager`` keeps for it
    (managers/event_manager.py: ``_reset_term_last_triggered_step_id`` / ``_reset_term_last_triggered_once``, ``_interval_term_time_left``)."""

    def __init__(self, name: str, term: dict, num_envs: int, robot, device):
        from .robots import resolve_matching_names

        self.name, self.cfg = name, term
        self.func = _func_name(term)
        if self.func not in _EVENT_OPS:
            raise NotImplementedError(f"event term '{name}': '{self.func}' has no kernel (known: {sorted(_EVENT_OPS)})")
        self.op = _EVENT_OPS[self.func]
        self.mode = term.get("mode")
        if self.mode not in ("reset", "interval"):
            raise ValueError(f"event term '{name}': mode '{self.mode}' is not run by the env step (reset / interval are)")
        if self.mode == "interval" and self.func not in _INTERVAL_OK:
            raise NotImplementedError(f"event term '{name}': '{self.func}' as an interval event is not supported")
        p = term.get("params", {}) or {}
        r = [0.0] * 24
        self.width = 0
        self.body_ids = None
        if self.func == "reset_root_state_uniform":
            r[:24] = _axis_ranges(p.get("pose_range")) + _axis_ranges(p.get("velocity_range"))
            self.width = 12
        elif self.func.startswith("reset_joints"):
            r[:4] = [float(x) for x in (*p["position_range"], *p["velocity_range"])]
            self.width = 2 * robot.num_joints
        elif self.func == "push_by_setting_velocity":
            r[:12] = _axis_ranges(p.get("velocity_range"))
            self.width = 6
        else:  # apply_external_force_torque: SceneEntityCfg body selection (scene_entity_cfg.py:220-250)
            r[:4] = [float(x) for x in (*p["force_range"], *p["torque_range"])]
            ent = p.get("asset_cfg") or {}
            names = ent.get("body_names") if isinstance(ent, dict) else getattr(ent, "body_names", None)
            ids = list(range(robot.num_bodies)) if names is None else resolve_matching_names(names, robot.body_names, False)[0]
            if len(ids) != robot.num_bodies:
                self.body_ids = torch.tensor(ids, dtype=torch.int32, device=device)
            self.width = 6 * len(ids)
        self.ranges = r
        self.is_global_time = bool(term.get("is_global_time", False))
        self.min_step_count_between_reset = int(term.get("min_step_count_between_reset", 0) or 0)
        self.interval_range_s = tuple(term.get("interval_range_s") or (0.0, 0.0))
        N = num_envs
        self.last_triggered_step = self.triggered_once = self.time_left = None
        if self.mode == "reset":
            self.last_triggered_step = torch.zeros(N, dtype=torch.int32, device=device)
            self.triggered_once = torch.zeros(N, dtype=torch.bool, device=device)
        else:
            if term.get("interval_range_s") is None:  # event_manager.py:_prepare_terms
                raise ValueError(f"Event term '{name}' has mode 'interval' but 'interval_range_s' is not specified.")
            lo, hi = self.interval_range_s
            if self.is_global_time:  # ONE timer; two slots (read [step & 1], write the other): both start at the first sample
                self.time_left = (torch.rand(1, device=device) * (hi - lo) + lo).repeat(2)
            else:
                self.time_left = torch.rand(N, device=device) * (hi - lo) + lo
        self.uniforms = None           # parity runs: (N, width) samples replacing the in-kernel draws
        self.interval_uniforms = None  # parity runs: (N) samples for the timer re-sampling


class EventManager:
    """``isaaclab.managers.EventManager`` surface (``active_terms``, ``available_modes``, ``reset``) over the device-side state of the
    reset / interval terms; ``apply`` is not a Python walk over id lists here but part of the env's ONE orchestration launch
    (``imx_reset_orchestrate``), which takes the step's reset mask and the timers and never reads anything back.  Startup-mode terms
    (PhysX materials, masses: simulator side) are listed in ``skipped_terms``, not run."""

    def __init__(self, events_cfg: dict, num_envs: int, robot, device, seed: int = 0):
        self.terms: list[EventTermState] = []
        self.skipped_terms: list[str] = []
        self.seed = int(seed)
        for name, term in (events_cfg or {}).items():
            if term is None:
                continue
            term = term if isinstance(term, dict) else term.to_dict()
            if term.get("mode") not in ("reset", "interval"):
                self.skipped_terms.append(name)
                continue
            self.terms.append(EventTermState(name, term, num_envs, robot, device))
        if len(self.terms) > _lib.ORCH_MAX_TERMS:
            raise NotImplementedError(f"{len(self.terms)} reset / interval event terms (at most {_lib.ORCH_MAX_TERMS})")

    @property
    def active_terms(self) -> dict:
        out: dict = {}
        for t in self.terms:
            out.setdefault(t.mode, []).append(t.name)
        return out

    @property
    def available_modes(self) -> list:
        return list(self.active_terms)

    def get_term(self, name: str) -> EventTermState:
        for t in self.terms:
            if t.name == name:
                return t
        raise ValueError(f"Event term '{name}' not found.")

    def reset(self, env_ids=None) -> dict:
        """event_manager.py:123-148: only CLASS terms are reset and only class-based interval terms get a new interval at an episode
        reset (the loop there walks ``_mode_class_term_cfgs``); the function terms handled here keep their timers.  Nothing to log."""
        return {}

    def fill(self, orch) -> None:
        orch.num_terms = len(self.terms)
        for i, t in enumerate(self.terms):
            T = orch.terms[i]
            T.op, T.mode = t.op, 0 if t.mode == "reset" else 1
            T.is_global_time, T.min_step_count_between_reset = int(t.is_global_time), t.min_step_count_between_reset
            T.interval_lo, T.interval_hi = float(t.interval_range_s[0]), float(t.interval_range_s[1])
            for k, v in enumerate(t.ranges):
                T.ranges[k] = v
            T.num_body_ids = 0 if t.body_ids is None else int(t.body_ids.numel())
            T.body_ids_d = _lib.ptr(t.body_ids)
            T.last_triggered_step_d, T.triggered_once_d = _lib.ptr(t.last_triggered_step), _lib.ptr(t.triggered_once)
            T.time_left_d = _lib.ptr(t.time_left)
            T.uniforms_d, T.interval_uniforms_d = _lib.ptr(t.uniforms), _lib.ptr(t.interval_uniforms)


class TerrainImporterState:
    """What ``TerrainImporter`` keeps for the terrain curriculum (terrains/terrain_importer.py:280-347): the grid of sub-terrain origins,
    every env's level (row) and type (column), and ``env_origins`` -- updated in place by the orchestration launch."""

    def __init__(self, terrain_origins: torch.Tensor, terrain_levels: torch.Tensor, terrain_types: torch.Tensor, size_x: float):
        self.terrain_origins = terrain_origins.contiguous().float()
        self.terrain_levels = terrain_levels.contiguous().long()
        self.terrain_types = terrain_types.contiguous().long()
        self.max_terrain_level = int(self.terrain_origins.shape[0])
        self.size_x = float(size_x)
        self.env_origins = self.terrain_origins[self.terrain_levels, self.terrain_types].clone()

    @classmethod
    def from_generator_cfg(cls, num_envs: int, generator_cfg: dict, device, max_init_terrain_level: int | None = None):
        """``_compute_env_origins_curriculum`` (:328-347) over the generator's grid: origin of sub-terrain (row, col) at the centre of its
        tile (terrain_generator.py: tiles of ``size`` laid out row-major around the world origin), height 0 (flat stand-in)."""
        R, C = int(generator_cfg["num_rows"]), int(generator_cfg["num_cols"])
        sx, sy = (float(v) for v in generator_cfg["size"])
        r, c = torch.meshgrid(torch.arange(R, dtype=torch.float32), torch.arange(C, dtype=torch.float32), indexing="ij")
        origins = torch.stack([(r + 0.5) * sx - R * sx * 0.5, (c + 0.5) * sy - C * sy * 0.5, torch.zeros(R, C)], dim=-1).to(device)
        max_init = R - 1 if max_init_terrain_level is None else min(int(max_init_terrain_level), R - 1)
        levels = torch.randint(0, max_init + 1, (num_envs,), device=device)
        types = torch.div(torch.arange(num_envs, device=device), (num_envs / C), rounding_mode="floor").to(torch.long)
        return cls(origins, levels, types, sx)


class CurriculumManager:
    """``isaaclab.managers.CurriculumManager`` surface for the one curriculum term with a kernel, ``terrain_levels_vel``; ``compute`` is
    part of the orchestration launch, ``reset`` reports the state the reference logs (curriculum_manager.py:95-118)."""

    def __init__(self, curriculum_cfg: dict, env):
        self._env = env
        self._term_names = []
        for name, term in (curriculum_cfg or {}).items():
            if term is None:
                continue
            fn = _func_name(term if isinstance(term, dict) else term.to_dict())
            if fn != "terrain_levels_vel":
                raise NotImplementedError(f"curriculum term '{name}': '{fn}' has no kernel (terrain_levels_vel has; modify_reward_weight is "
                                          "RewardManager.set_term_cfg on the host)")
            self._term_names.append(name)

    @property
    def active_terms(self) -> list:
        return list(self._term_names)

    def reset(self, env_ids=None) -> dict:
        ti = self._env.terrain_importer
        return {f"Curriculum/{n}": torch.mean(ti.terrain_levels.float()) for n in self._term_names}
