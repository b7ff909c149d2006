"""``ManagerBasedRLEnv`` on libimx: the reference's gym surface (isaaclab/envs/manager_based_rl_env.py:26-392,
isaaclab/envs/manager_based_env.py:264-315) with the post-physics step executed by three HIP kernels.

    step(action):  imx_action_process -> [physics = StateFeed.advance()] -> imx_terminations_rewards
                   -> imx_observations            (no host synchronisation anywhere on the path)

PhysX / InteractiveScene / Event- / Curriculum- / CommandManager are out of scope: their
outputs arrive as tensors from a :class:`~isaaclab_amd.state_feed.StateFeed`.  The manager objects below keep the
reference's public attribute names so that user code (`env.reward_manager._episode_sums[...]`,
`env.termination_manager.time_outs`, `env.action_manager.action` ...) keeps working; they are views over the
buffers the kernels write.
"""

from __future__ import annotations

import ctypes
import importlib
import json
import math
import os
from collections.abc import Sequence
from typing import Any

import numpy as np
import torch

from . import _lib
from ._lib import ImxBuffers, ImxState, check, lib
from .plan import Plan, compile_plan
from .plan import func_name as func_name_of
from .robots import ROBOTS, RobotSpec
from .state_feed import StateFeed

_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def load_task_cfg(task: str) -> dict:
    """Config fixture captured from the reference's gym registry entry ``task`` (``cfg.to_dict()`` form)."""
    path = os.path.join(_CFG_DIR, task + ".json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"no config fixture for task '{task}' under {_CFG_DIR}")
    with open(path) as f:
        fx = json.load(f)
    side = os.path.join(_CFG_DIR, task + ".managers.json")  # events / curriculum / robot init state of the task, dumped separately
    if os.path.exists(side):
        with open(side) as f:
            extra = json.load(f)
        env = fx["env"]
        env.setdefault("events", extra.get("events"))
        env.setdefault("curriculum", extra.get("curriculum"))
        for k, v in (extra.get("scene") or {}).items():
            env["scene"].setdefault(k, v)
    return fx


def _string_to_callable(name: str):
    """isaaclab/utils/string.py:138-176"""
    mod, _, attr = name.partition(":")
    obj = importlib.import_module(mod)
    for part in attr.split("."):
        obj = getattr(obj, part)
    if not callable(obj):
        raise AttributeError(f"The imported object is not callable: '{name}'")
    return obj


class TerrainMesh:
    """Device mesh + grid for the ray-caster (``imx_mesh_create``)."""

    def __init__(self, vertices: np.ndarray, triangles: np.ndarray, cell_size: float = 0.0):
        v = np.ascontiguousarray(vertices, np.float32)
        t = np.ascontiguousarray(triangles, np.uint32)
        if v.ndim != 2 or v.shape[1] != 3 or t.ndim != 2 or t.shape[1] != 3:
            raise ValueError("vertices must be (V,3) float and triangles (F,3) integer arrays")
        self._h = ctypes.c_void_p()
        check(lib().imx_mesh_create(v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], float(cell_size),
                                    ctypes.byref(self._h)))
        info = (ctypes.c_int64 * 8)()
        check(lib().imx_mesh_info(self._h, info))
        self.nx, self.ny, self.num_triangles, self.num_refs = (int(info[i]) for i in range(4))
        self.max_refs, self.num_flat_cells = int(info[4]) & 0xFFFFFFFF, int(info[4]) >> 32  # FLAT: general cells answered by their descriptor
        self.num_lattice_cells, self.num_general_cells = int(info[5]), int(info[6])
        self.num_vertices = v.shape[0]

    @property
    def handle(self):
        return self._h

    def raycast(self, ray_starts: torch.Tensor, ray_directions: torch.Tensor, max_dist: float = 1e6,
                return_distance: bool = False, return_face_id: bool = False):
        """``raycast_mesh`` (isaaclab/utils/warp/ops.py:24-127): hits (+inf on miss) [, distance, face id]."""
        shape = ray_starts.shape
        s = ray_starts.reshape(-1, 3).contiguous().float()
        d = ray_directions.reshape(-1, 3).contiguous().float()
        n = s.shape[0]
        hits = torch.empty_like(s)
        dist = torch.empty(n, device=s.device) if return_distance else None
        face = torch.empty(n, dtype=torch.int32, device=s.device) if return_face_id else None
        check(lib().imx_raycast(self._h, s.data_ptr(), d.data_ptr(), n, float(max_dist), hits.data_ptr(),
                                _lib.ptr(dist), _lib.ptr(face), _lib.current_stream(s.device)))
        return (hits.view(shape), None if dist is None else dist.view(shape[:-1]), None,
                None if face is None else face.view(shape[:-1]))

    def __del__(self):
        try:
            if self._h:
                lib().imx_mesh_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------- term contract
class ManagerTermBase:
    """The class form of a manager term (managers/manager_base.py:28-115): built once as ``cls(cfg=term_cfg, env=env)``, called every step
    as ``term(env, **params)``, ``reset(env_ids)`` for the envs of a reset.  Stand-in for users without ``isaaclab`` installed; a subclass of
    the reference's own ``isaaclab.managers.ManagerTermBase`` is handled the same way (any class given as ``func`` is)."""

    def __init__(self, cfg, env):
        self.cfg = cfg
        self._env = env

    @property
    def num_envs(self) -> int:
        return self._env.num_envs

    @property
    def device(self):
        return self._env.device

    def reset(self, env_ids=None) -> None:
        pass

    def __call__(self, *args):
        raise NotImplementedError("The method '__call__' should be implemented by the subclass.")


# ---------------------------------------------------------------------------------------------------- manager views
class TermCfgView:
    """What ``get_term_cfg`` hands out (RewardTermCfg / TerminationTermCfg surface: ``func``, ``params``, ``weight`` /
    ``time_out``): a mutable copy of the term's cfg entry.  Changing it has no effect until it goes back through ``set_term_cfg``,
    as in the reference where the caller mutates the cfg object and re-installs it (envs/mdp/curriculums.py:32-36)."""

    def __init__(self, entry: dict):
        import copy

        self.__dict__.update(copy.deepcopy(entry))

    def to_dict(self) -> dict:
        return dict(self.__dict__)


def _cfg_entry(cfg: Any) -> dict:
    if isinstance(cfg, dict):
        return dict(cfg)
    if hasattr(cfg, "to_dict"):
        return dict(cfg.to_dict())
    return dict(vars(cfg))


class _ActionTermView:
    def __init__(self, env, name, col0, dim):
        self._env, self.name, self._c0, self.action_dim = env, name, col0, dim

    @property
    def raw_actions(self):
        return self._env._action[:, self._c0:self._c0 + self.action_dim]

    @property
    def processed_actions(self):
        return self._env._processed_action[:, self._c0:self._c0 + self.action_dim]


class ActionManager:
    """View with the public surface of isaaclab/managers/action_manager.py:228-359."""

    def __init__(self, env: "ManagerBasedRLEnv"):
        self._env = env
        self._terms = {}
        c = 0
        for t in env.plan.action_terms:
            self._terms[t.name] = _ActionTermView(env, t.name, c, t.dim)
            c += t.dim

    @property
    def active_terms(self) -> list[str]:
        return list(self._terms)

    @property
    def total_action_dim(self) -> int:
        return self._env.plan.action_dim

    @property
    def action_term_dim(self) -> list[int]:
        return [t.action_dim for t in self._terms.values()]

    @property
    def action(self) -> torch.Tensor:
        return self._env._action

    @property
    def prev_action(self) -> torch.Tensor:
        return self._env._prev_action

    def get_term(self, name: str):
        return self._terms[name]

    def process_action(self, action: torch.Tensor):
        self._env._process_action(action)

    def apply_action(self):  # the processed targets would go to PhysX here
        pass

    def reset(self, env_ids=None) -> dict:
        ids = slice(None) if env_ids is None else env_ids
        self._env._prev_action[ids] = 0.0
        self._env._action[ids] = 0.0
        # EMAJointPositionToLimitsAction.reset (joint_actions_to_limits.py:208-217): the moving average restarts from the joint positions
        # (inside env.step() the action kernel does this itself for the envs the step kernel reset)
        c = 0
        for t in self._env.plan.action_terms:
            if t.func.rsplit(":", 1)[-1].rsplit(".", 1)[-1] == "EMAJointPositionToLimitsAction":
                from .robots import resolve_matching_names

                jids = resolve_matching_names(t.params["joint_names"], self._env.plan.robot.joint_names, bool(t.params.get("preserve_order")))[0]
                rows = torch.arange(self._env.num_envs, device=self._env.device)[ids]
                self._env._processed_action[rows.unsqueeze(1), torch.arange(c, c + t.dim, device=self._env.device)] = \
                    self._env.feed["joint_pos"][rows][:, jids]
            c += t.dim
        return {}


class ObservationManager:
    """isaaclab/managers/observation_manager.py:177-335 surface; every (concatenated) group of the cfg is filled by the one fused
    launch of ``imx_observations``."""

    def __init__(self, env: "ManagerBasedRLEnv"):
        self._env = env

    @property
    def _groups(self):
        return self._env.plan.obs_groups

    @property
    def active_terms(self):
        return {g.name: [t.name for t in g.terms] for g in self._groups}

    @property
    def group_obs_dim(self):
        """observation_manager.py:85-101: the concatenated shape -- the reference SUMS the term shapes element-wise, so terms that keep
        their history axis add up in that axis too, (H, d1) + (H, d2) -> (2H, d1 + d2); mirrored -- or the list of term shapes."""
        out = {}
        for g in self._groups:
            if g.concatenate:
                out[g.name] = tuple(int(sum(d[i] for d in g.term_dims)) for i in range(len(g.term_dims[0])))
            else:
                out[g.name] = list(g.term_dims)
        return out

    @property
    def group_obs_term_dim(self):
        return {g.name: list(g.term_dims) for g in self._groups}

    @property
    def group_obs_concatenate(self):
        return {g.name: g.concatenate for g in self._groups}

    def shaped(self) -> dict:
        """The groups as the reference hands them out (observation_manager.py:320-335).  A concatenated group of one-dimensional terms
        IS its fused row; a group of terms that keep their history axis is the concatenation of the (N, H, d) windows along the last
        axis; ``concatenate_terms=False`` gives the dict of terms -- views of the row the kernel filled, no extra launch."""
        out = {}
        for g, row in zip(self._groups, self._env._obs_groups):
            if g.is_flat:
                out[g.name] = row
                continue
            terms, off = {}, 0
            for t, shape, w in zip(g.terms, g.term_dims, g.term_widths):
                v = row[:, off:off + w]
                terms[t.name] = v.unflatten(1, shape) if len(shape) > 1 else v
                off += w
            out[g.name] = torch.cat(list(terms.values()), dim=-1) if g.concatenate else terms
        return out

    def compute(self) -> dict:
        """observation_manager.py:238-258: every group (one launch fills them all)."""
        self._env._compute_observations()
        return self.shaped()

    def compute_group(self, group_name: str):
        if group_name not in self._env.obs_buf:  # observation_manager.py:293-297
            raise ValueError(f"Unable to find the group '{group_name}' in the observation manager."
                             f" Available groups are: {list(self._env.obs_buf)}")
        self._env._compute_observations()
        return self.shaped()[group_name]

    def reset(self, env_ids=None) -> dict:
        return {}


class RewardManager:
    """isaaclab/managers/reward_manager.py:91-209 surface; buffers are written by imx_terminations_rewards."""

    def __init__(self, env: "ManagerBasedRLEnv"):
        self._env = env
        self._term_names = [t.name for t in env.plan.reward_terms]
        self._episode_sums = {n: env._episode_sums[i] for i, n in enumerate(self._term_names)}
        self._step_reward = env._step_reward
        self._reward_buf = env._reward_buf

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    def compute(self, dt: float | None = None) -> torch.Tensor:
        """The reward is produced together with the terminations by ``env.step``; this returns that buffer."""
        return self._reward_buf

    def get_term_cfg(self, term_name: str):
        """reward_manager.py:178-193"""
        if term_name not in self._term_names:
            raise ValueError(f"Reward term '{term_name}' not found.")
        return TermCfgView(self._env._cfg_dict["rewards"][term_name])

    def set_term_cfg(self, term_name: str, cfg):
        """reward_manager.py:163-176.  The term tables live on the device: the changed cfg is recompiled and copied over them in
        place (``imx_plan_update``, stream-ordered; graphs captured on this env stay valid).  A cfg equal to the installed one --
        ``modify_reward_weight`` re-installs its weight at every reset -- costs a dict comparison."""
        if term_name not in self._term_names:
            raise ValueError(f"Reward term '{term_name}' not found.")
        self._env._install_term_cfg("rewards", term_name, _cfg_entry(cfg))

    def reset(self, env_ids=None) -> dict:
        """Episode_Reward/<term> = mean(episode_sum[ids]) / max_episode_length_s, then zero (reward_manager.py:100-126)."""
        ids = slice(None) if env_ids is None else env_ids
        extras = {}
        for i, key in enumerate(self._term_names):
            extras["Episode_Reward/" + key] = torch.mean(self._env._episode_sums[i][ids]) / self._env.max_episode_length_s
            self._env._episode_sums[i][ids] = 0.0
        return extras

    def get_active_iterable_terms(self, env_idx: int):
        return [(n, [self._step_reward[env_idx, i].cpu().item()]) for i, n in enumerate(self._term_names)]


class TerminationManager:
    """isaaclab/managers/termination_manager.py:94-185 surface."""

    def __init__(self, env: "ManagerBasedRLEnv"):
        self._env = env
        self._term_names = [t.name for t in env.plan.termination_terms]
        self._term_dones = {n: env._term_dones[i] for i, n in enumerate(self._term_names)}

    @property
    def active_terms(self) -> list[str]:
        return self._term_names

    @property
    def dones(self) -> torch.Tensor:
        return self._env.reset_buf

    @property
    def time_outs(self) -> torch.Tensor:
        return self._env.reset_time_outs

    @property
    def terminated(self) -> torch.Tensor:
        return self._env.reset_terminated

    def compute(self) -> torch.Tensor:
        return self._env.reset_buf

    def get_term(self, name: str) -> torch.Tensor:
        return self._term_dones[name]

    def find_terms(self, name_keys):
        from .robots import resolve_matching_names

        return resolve_matching_names(name_keys, self._term_names)[1]

    def get_term_cfg(self, term_name: str):
        """termination_manager.py:222-237"""
        if term_name not in self._term_names:
            raise ValueError(f"Termination term '{term_name}' not found.")
        return TermCfgView(self._env._cfg_dict["terminations"][term_name])

    def set_term_cfg(self, term_name: str, cfg):
        """termination_manager.py:207-220 (see RewardManager.set_term_cfg)"""
        if term_name not in self._term_names:
            raise ValueError(f"Termination term '{term_name}' not found.")
        self._env._install_term_cfg("terminations", term_name, _cfg_entry(cfg))

    def reset(self, env_ids=None) -> dict:
        ids = slice(None) if env_ids is None else env_ids
        return {"Episode_Termination/" + k: torch.count_nonzero(v[ids]).item() for k, v in self._term_dones.items()}


class CommandManager:
    """Commands come from the feed (UniformVelocityCommand is a SURVEY 8f 'next' row)."""

    def __init__(self, env):
        self._env = env

    def get_command(self, name: str) -> torch.Tensor:
        if self._env.command_term is not None:
            return self._env.command_term.command
        return self._env.feed["command"]

    def compute(self, dt: float):
        pass

    def reset(self, env_ids=None) -> dict:
        return {}


class _FeedData:
    """``asset.data`` / ``sensor.data`` for Python fallback terms: feed tensors + libimx-derived root-frame vectors."""

    def __init__(self, env):
        self._env = env

    def __getattr__(self, name):
        env = self.__dict__["_env"]
        if name in ("root_lin_vel_b", "root_ang_vel_b", "projected_gravity_b"):
            return env._root_frame()[name]
        if name == "root_link_quat_w":
            name = "root_quat_w"
        try:
            return env.feed[name]
        except KeyError:
            raise AttributeError(name) from None


class _Entity:
    def __init__(self, env, names_joint, names_body):
        self.data = _FeedData(env)
        self.joint_names, self.body_names = list(names_joint), list(names_body)
        self.num_joints, self.num_bodies = len(names_joint), len(names_body)

    def find_joints(self, keys, joint_subset=None, preserve_order=False):
        from .robots import resolve_matching_names

        return resolve_matching_names(keys, self.joint_names, preserve_order)

    def find_bodies(self, keys, preserve_order=False):
        from .robots import resolve_matching_names

        return resolve_matching_names(keys, self.body_names, preserve_order)


class _SceneEntityView:
    """``SceneEntityCfg`` as a Python-evaluated term sees it after ``resolve()`` (managers/scene_entity_cfg.py:112-250): ``name``,
    ``joint_names`` / ``body_names`` and the resolved ``joint_ids`` / ``body_ids`` (``slice(None)`` when every one is selected)."""

    def __init__(self, ent: dict, compiler):
        self.name = ent.get("name")
        self.joint_names, self.body_names = ent.get("joint_names"), ent.get("body_names")
        self.preserve_order = bool(ent.get("preserve_order"))
        for kind in ("joint", "body"):
            ids = slice(None)
            try:
                n_all = len(compiler._entity_names(self.name, kind))
                r = compiler.resolve_ids(ent, kind)
                ids = slice(None) if (ent.get(f"{kind}_names") is None and len(r) == n_all) else r
            except ValueError:
                pass  # entity without such a name table (e.g. the height scanner)
            setattr(self, f"{kind}_ids", ids)


def _looks_like_scene_entity(v) -> bool:
    return isinstance(v, dict) and "name" in v and any(k in v for k in ("joint_names", "body_names", "joint_ids", "body_ids"))


class _Scene:
    def __init__(self, env):
        r = env.plan.robot
        self._e = {"robot": _Entity(env, r.joint_names, r.body_names)}
        self.sensors = {"contact_forces": _Entity(env, [], r.body_names)}
        self._e.update(self.sensors)
        self._env = env
        self.num_envs = env.num_envs

    @property
    def env_origins(self):
        return self._env.feed["env_origins"]

    def keys(self):
        return list(self._e)

    def __getitem__(self, k):
        return self._e[k]


def _seed_process(seed: int = -1) -> int:
    """``ManagerBasedEnv.seed`` (envs/manager_based_env.py:425-443, a ``@staticmethod``): torch / numpy / random seeding."""
    import random

    torch.manual_seed(seed)
    np.random.seed(seed % (2 ** 32))
    random.seed(seed)
    return seed


def _seed_env(env, seed: int = -1) -> int:
    """The same on an instance -- and the seeds of the in-kernel generators (observation noise, sensor drift; command, reset-event and
    interval-event draws when the env owns those producers), so that "same seed, same rollout"
    (isaaclab_tasks/test/test_environment_determinism.py:57-66) holds for seeds given after construction too."""
    _seed_process(seed)
    if getattr(env, "_counters", None) is not None:
        env.noise_seed = int(seed) & 0xFFFFFFFFFFFF  # streams are keyed by (seed, step counter, env, column)
        for name in ("command_term", "reset_events", "event_manager"):
            prod = getattr(env, name, None)
            if prod is not None and hasattr(prod, "seed"):
                prod.seed = env.noise_seed
    return seed


class _SeedMethod:
    """Descriptor: ``ManagerBasedRLEnv.seed(42)`` works on the class as in the reference, ``env.seed(42)`` also reaches the kernels."""

    def __get__(self, obj, cls=None):
        if obj is None:
            return _seed_process
        import functools

        return functools.partial(_seed_env, obj)


# ---------------------------------------------------------------------------------------------------- the env
class ManagerBasedRLEnv:
    """Drop-in for ``isaaclab.envs.ManagerBasedRLEnv`` on the fused HIP path.

    Args:
        cfg: a reference ``ManagerBasedRLEnvCfg`` instance, its ``to_dict()`` form, a fixture dict from
            :func:`load_task_cfg`, or a task id string (e.g. ``"Isaac-Velocity-Rough-Anymal-C-v0"``).
        render_mode: kept for signature compatibility (rendering is out of scope).
        state_feed: the :class:`StateFeed` standing in for PhysX; synthetic one is generated when omitted.
        terrain: ``(vertices, triangles)`` or :class:`TerrainMesh` for configs with a height scanner.
    """

    metadata = {"render_modes": [None]}
    is_vector_env = True

    def __init__(self, cfg: Any, render_mode: str | None = None, *, state_feed: StateFeed | None = None,
                 robot: RobotSpec | str | None = None, terrain=None, num_envs: int | None = None,
                 device: str | torch.device | None = None, seed: int | None = None, noise_seed: int = 0,
                 terrain_cell: float = 0.0, use_command_term: bool = False, use_contact_sensor: bool = False,
                 events_cfg: dict | bool | None = None, use_curriculum: bool = False, terrain_importer=None, own_managers: bool = False,
                 **kwargs):
        """``own_managers=True``: the env runs its cfg's EventManager (reset / interval terms), CommandManager (UniformVelocityCommand)
        and CurriculumManager (terrain_levels_vel) itself -- ``_reset_idx`` + the command / interval updates of ``step`` as ONE
        orchestration launch (``imx_reset_orchestrate``) -- instead of taking commands from the feed; the single switches
        (``events_cfg`` = a dict or True for the cfg's own, ``use_command_term``, ``use_curriculum``) select parts of it.
        ``terrain_importer``: an ``events.TerrainImporterState`` (default: built from the cfg's terrain generator grid)."""
        if own_managers:
            ec = cfg.get("env", cfg) if isinstance(cfg, dict) else None
            ec = ec if ec is not None else (load_task_cfg(cfg)["env"] if isinstance(cfg, str) else cfg.to_dict())
            use_command_term = use_command_term or bool(ec.get("commands"))
            events_cfg = events_cfg if events_cfg is not None else (True if ec.get("events") else None)
            use_curriculum = use_curriculum or any(v is not None for v in (ec.get("curriculum") or {}).values())
        if isinstance(cfg, str):
            cfg = load_task_cfg(cfg)
        self.cfg = cfg
        env_cfg = cfg
        robot_name = robot
        if isinstance(cfg, dict) and "env" in cfg and "task" in cfg:  # fixture wrapper
            env_cfg = cfg["env"]
            robot_name = robot_name or cfg.get("robot")
            self.agent_cfg = cfg.get("agent")
        if isinstance(robot_name, str):
            robot_name = ROBOTS[robot_name]
        if robot_name is None:
            if state_feed is None:
                raise ValueError("pass robot= (RobotSpec or name) or a state_feed when cfg is not a task fixture")
            robot_name = state_feed.robot
        self.render_mode = render_mode
        import copy

        # the env's own copy of the cfg in to_dict() form: set_term_cfg edits and recompiles it
        self._cfg_dict = copy.deepcopy(env_cfg if isinstance(env_cfg, dict) else env_cfg.to_dict())
        self._robot = robot_name
        self.plan: Plan = compile_plan(self._cfg_dict, robot_name)
        plan = self.plan
        env_dict = self._cfg_dict
        if device is None:
            device = state_feed.device if state_feed is not None else ("cuda:0" if torch.cuda.is_available() else None)
        if device is None or torch.device(device).type != "cuda":
            raise _lib.ImxError("ManagerBasedRLEnv runs on an MI355X through libimx; no GPU device is available "
                                f"(device={device}). There is no CPU fallback.")
        self.device = torch.device(device)
        self._lib = lib()
        N = num_envs or (state_feed.num_envs if state_feed is not None else int(env_dict["scene"]["num_envs"]))
        self.num_envs = N
        extent = None
        self.terrain: TerrainMesh | None = None
        if plan.num_rays > 0:
            if terrain is None:
                raise ValueError("this config has a height scanner: pass terrain=(vertices, triangles)")
            if isinstance(terrain, TerrainMesh):
                self.terrain = terrain
            else:
                verts, tris = terrain[0], terrain[1]
                self.terrain = TerrainMesh(verts, tris, terrain_cell)
                v = np.asarray(verts)
                extent = (float(np.abs(v[:, 0]).max()) - 2.0, float(np.abs(v[:, 1]).max()) - 2.0)
        if state_feed is None:
            state_feed = StateFeed(plan.robot, N, self.device, seed=42 if seed is None else seed, extent_xy=extent,
                                   history=plan.history, gravity=tuple(env_dict["sim"].get("gravity", (0, 0, -9.81))))
        if state_feed.num_envs != N or torch.device(state_feed.device) != self.device:
            raise ValueError("state_feed has a different num_envs/device than the env")
        self.feed = state_feed
        self.step_dt = plan.step_dt
        self.physics_dt = float(env_dict["sim"]["dt"])
        self.max_episode_length = plan.max_episode_length
        self.max_episode_length_s = plan.max_episode_length_s
        self.common_step_counter = 0
        self._sim_step_counter = 0
        self._noise_seed = int(noise_seed)  # the property below also leaves it in counters[4..5] (sensor-drift stream of the step kernel)
        self.clip_actions: float | None = None  # set by RslRlVecEnvWrapper: fused into imx_action_process
        self.extras: dict = {}

        # ---- persistent buffers (manager state); pointers stay fixed so steps can be captured in a hipGraph
        dev, K, NT = self.device, max(len(plan.reward_terms), 1), max(len(plan.termination_terms), 1)
        A, D = plan.action_dim, plan.obs_dim
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self._episode_length_buf = z(N, dtype=torch.long)
        self._action, self._prev_action, self._processed_action = z(N, max(A, 1)), z(N, max(A, 1)), z(N, max(A, 1))
        self._reward_buf = z(N)
        self._episode_sums = z(K, N)
        self._step_reward = z(N, K)
        self._term_dones = z(NT, N, dtype=torch.bool)
        self.reset_terminated, self.reset_time_outs, self.reset_buf = (z(N, dtype=torch.bool) for _ in range(3))
        self._reset_env_ids = z(N, dtype=torch.long)
        self._counters = z(8, dtype=torch.int32)
        self.noise_seed = self._noise_seed
        self._log_out = z(K + NT + 1 + 3)  # + the orchestration's entries: Metrics/<command>/error_vel_xy|yaw, Curriculum/terrain_levels
        self._obs_groups = [z(N, max(g.dim, 1)) for g in plan.obs_groups]  # one (N, D_g) tensor per observation group
        self._obs = self._obs_groups[0]
        self._ext_reward = z(N, plan.n_ext_rew) if plan.n_ext_rew else None
        self._ext_term = z(N, plan.n_ext_term, dtype=torch.bool) if plan.n_ext_term else None
        self._ext_obs = z(N, plan.n_ext_obs) if plan.n_ext_obs else None
        self._noise_u: torch.Tensor | None = None  # parity mode: uniforms replacing torch.rand_like
        self.materialize_ray_hits = False
        self._ray_hits = None
        self.reward_buf = self._reward_buf
        self.obs_buf = {g.name: t for g, t in zip(plan.obs_groups, self._obs_groups)}
        # height scanner as a SensorBase (update_period gating, drift): per env {timestamp, last update, drift xyz, data.pos_w z,
        # outdated, step stamp} + the hit heights of envs that skip an update; sensors start outdated (sensor_base.py:_initialize_impl)
        self._scan_state = self._scan_hit_z = None
        self._scan_drift_feed: torch.Tensor | None = None  # parity runs: (N,3) drift values taken at a sensor reset
        self.defer_step_tail = True  # the step's tail (ordered reset ids, Episode_* log) rides in the observation launch (include/imx.h)
        self.scanner_keep_all_hits = False  # keep every env's hit heights (set before overwriting the sensor timestamps by hand)
        if plan.scan_stateful:
            self._scan_state = z(N, 8)
            self._scan_state[:, 6] = 1.0
            self._scan_hit_z = z(N, plan.num_rays)

        blob = np.ascontiguousarray(plan.blob, np.int32)
        self._plan_h = ctypes.c_void_p()
        check(self._lib.imx_plan_create(blob.ctypes.data, blob.size, ctypes.byref(self._plan_h)))
        self._scratch = torch.zeros(int(self._lib.imx_plan_scratch_bytes(self._plan_h, N)), dtype=torch.uint8, device=dev)
        # DigitalFilter / Integrator state of observation modifiers (utils/modifiers/modifier.py), zeroed per env on reset by k_obs
        self._mod_state = torch.zeros(N, plan.mod_state_dim, device=dev) if plan.mod_state_dim > 0 else None
        self._bufs = ImxBuffers(
            episode_length_buf=self._episode_length_buf.data_ptr(), action=self._action.data_ptr(),
            prev_action=self._prev_action.data_ptr(), processed_action=self._processed_action.data_ptr(),
            reward_buf=self._reward_buf.data_ptr(), episode_sums=self._episode_sums.data_ptr(),
            step_reward=self._step_reward.data_ptr(), term_dones=self._term_dones.data_ptr(),
            terminated=self.reset_terminated.data_ptr(), truncated=self.reset_time_outs.data_ptr(),
            reset_buf=self.reset_buf.data_ptr(), reset_env_ids=self._reset_env_ids.data_ptr(),
            counters=self._counters.data_ptr(), log_out=self._log_out.data_ptr(), obs=self._obs.data_ptr(),
            scratch=self._scratch.data_ptr(), mod_state=self._mod_state.data_ptr() if self._mod_state is not None else None,
            scan_state=_lib.ptr(self._scan_state), scan_hit_z=_lib.ptr(self._scan_hit_z),
            **{f"obs_extra{g}": self._obs_groups[g].data_ptr() for g in range(1, len(self._obs_groups))})
        self._state_cache: dict[int, ImxState] = {}
        self._root_cache = None

        self.action_manager = ActionManager(self)
        self.observation_manager = ObservationManager(self)
        self.reward_manager = RewardManager(self)
        self.termination_manager = TerminationManager(self)
        self.command_manager = CommandManager(self)
        # -- optional producers run by the env itself instead of arriving through the feed
        self.command_term = None
        if use_command_term:
            from .producers import UniformVelocityCommand

            cmds = (env_dict.get("commands") or {})
            if not cmds:
                raise ValueError("use_command_term=True but the env cfg has no command terms")
            self.command_term = UniformVelocityCommand(next(iter(cmds.values())), N, plan.step_dt, self.device, seed=noise_seed)
        self.contact_sensor = None
        if use_contact_sensor:
            from .producers import ContactSensorState

            cs = env_dict["scene"].get("contact_forces")
            if cs is None:
                raise ValueError("use_contact_sensor=True but the scene has no contact_forces sensor")
            self.contact_sensor = ContactSensorState(N, plan.num_bodies, int(cs.get("history_length", 0)),
                                                     bool(cs.get("track_air_time", False)), float(cs.get("update_period", 0.0)),
                                                     float(cs.get("force_threshold", 1.0)), self.device)
        # -- the reset / interval orchestration: EventManager.apply("reset" | "interval"), the terrain curriculum, the
        # command term's reset / compute and scene.reset as ONE launch on the step kernel's reset mask.  What the reference hands to
        # write_root_pose_to_sim / write_root_velocity_to_sim / write_joint_state_to_sim / set_external_force_and_torque lands in
        # ``sim_writes`` (the simulator itself is out of scope; the feed keeps playing its snapshots)
        self.event_manager = self.curriculum_manager = self.terrain_importer = None
        self.actuator_net = None  # an env-owned ActuatorNetLSTM (attach_actuator): its state restarts with the env (scene.reset)
        self.articulation = None  # an env-owned producers.ArticulationRootState (use_articulation_update=True)
        if kwargs.pop("use_articulation_update", False):
            from .producers import ArticulationRootState

            self.articulation = ArticulationRootState(N, plan.num_joints, self.device)
        self._orch = None
        J = plan.num_joints
        if events_cfg:
            from .events import EventManager

            ev = env_dict.get("events") if events_cfg is True else events_cfg
            if not ev:
                raise ValueError("events_cfg=True but the env cfg has no events")
            self.event_manager = EventManager(ev, N, plan.robot, self.device, seed=noise_seed)
        if use_curriculum:
            from .events import CurriculumManager, TerrainImporterState

            if terrain_importer is None:
                tg = ((env_dict.get("scene") or {}).get("terrain") or {}).get("terrain_generator")
                if not tg:
                    raise ValueError("use_curriculum=True needs a terrain generator grid in the cfg or terrain_importer=")
                terrain_importer = TerrainImporterState.from_generator_cfg(N, tg, self.device)
            self.terrain_importer = terrain_importer
            self.curriculum_manager = CurriculumManager(env_dict.get("curriculum") or {}, self)
            if self.command_term is None:
                raise ValueError("the terrain curriculum reads the env's own velocity command: use_command_term=True")
        if self.event_manager is not None or self.command_term is not None or self.curriculum_manager is not None:
            init = ((env_dict.get("scene") or {}).get("robot") or {}).get("init_state") or {}
            drs = torch.zeros(N, 13, device=self.device)
            drs[:, 0:3] = torch.tensor(init.get("pos", (0.0, 0.0, 0.6)), device=self.device)
            drs[:, 3:7] = torch.tensor(init.get("rot", (1.0, 0.0, 0.0, 0.0)), device=self.device)
            drs[:, 7:10] = torch.tensor(init.get("lin_vel", (0.0, 0.0, 0.0)), device=self.device)
            drs[:, 10:13] = torch.tensor(init.get("ang_vel", (0.0, 0.0, 0.0)), device=self.device)
            self.default_root_state = drs
            NB = plan.robot.num_bodies
            self.sim_writes = {"root_pose": torch.zeros(N, 7, device=self.device), "root_vel": torch.zeros(N, 6, device=self.device),
                               "joint_pos": torch.zeros(N, J, device=self.device), "joint_vel": torch.zeros(N, J, device=self.device),
                               "ext_force": torch.zeros(N, NB, 3, device=self.device), "ext_torque": torch.zeros(N, NB, 3, device=self.device)}
            self._ev_part = torch.zeros(int(self._lib.imx_orch_part_floats(N)), device=self.device)
            self._orch_draws = {}  # parity runs: "command" -> (2,N,7) uniforms, "rand_levels" -> (N) int64
            self.defer_step_tail = True
        self.scene = _Scene(self)
        self._class_terms: list = []  # instances of class-based Python-evaluated terms, in construction order
        self._ext_funcs = {
            "rew": [(t, self._resolve_ext(t)) for t in plan.reward_terms if t.external is not None],
            "term": [(t, self._resolve_ext(t)) for t in plan.termination_terms if t.external is not None],
            "obs": [(t, self._resolve_ext(t)) for g in plan.obs_groups for t in g.terms if t.external is not None],
        }
        names_r = [t.name for t in plan.reward_terms]
        names_t = [t.name for t in plan.termination_terms]
        self._log_index = {"Episode_Reward/" + n: i for i, n in enumerate(names_r)}
        self._log_index.update({"Episode_Termination/" + n: len(names_r) + i for i, n in enumerate(names_t)})
        base = len(names_r) + len(names_t) + 1  # (the slot before holds the reset count)
        ev_flags = 0
        if self.command_term is not None:  # CommandManager.reset (command_manager.py:340-358): "Metrics/{term}/{metric}"
            cname = next(iter(env_dict.get("commands") or {"base_velocity": None}))
            self._log_index[f"Metrics/{cname}/error_vel_xy"] = base
            self._log_index[f"Metrics/{cname}/error_vel_yaw"] = base + 1
            ev_flags |= 1
        if self.curriculum_manager is not None:
            for n in self.curriculum_manager.active_terms:  # curriculum_manager.py:95-118
                self._log_index[f"Curriculum/{n}"] = base + 2
            ev_flags |= 2
        self._log_views = {k: self._log_out[i] for k, i in self._log_index.items()}
        if getattr(self, "_ev_part", None) is not None:
            self._bufs.ev_part = self._ev_part.data_ptr()
            self._bufs.ev_flags = ev_flags
        self._configure_gym_env_spaces()
        if seed is not None:
            self.seed(seed)

    # ---- gym-ish properties ------------------------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    @property
    def episode_length_buf(self) -> torch.Tensor:
        return self._episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value: torch.Tensor):
        # RSL-RL's init_at_random_ep_len assigns a new tensor (vecenv_wrapper.py:144-156); keep the pointer stable
        self._episode_length_buf.copy_(value.to(self.device, torch.long))

    @property
    def noise_seed(self) -> int:
        """Seed of the in-kernel generators (observation noise: an ``imx_observations`` argument; sensor drift: ``counters[4..5]``)."""
        return self._noise_seed

    @noise_seed.setter
    def noise_seed(self, value: int):
        self._noise_seed = int(value)
        if getattr(self, "_counters", None) is not None:
            lo, hi = self._noise_seed & 0xFFFFFFFF, (self._noise_seed >> 32) & 0xFFFFFFFF
            words = torch.tensor([lo, hi], dtype=torch.int64).to(torch.int32)  # two's-complement wrap
            self._counters[4:6].copy_(words)

    @property
    def reset_env_ids(self) -> torch.Tensor:
        """Ascending ids of the envs reset in the last step (host sync: reads the device-side count)."""
        return self._reset_env_ids[: int(self._counters[0].item())]

    seed = _SeedMethod()  # callable on the class (the reference's @staticmethod) and on an instance (also reseeds the kernels)

    def _configure_gym_env_spaces(self):
        """manager_based_rl_env.py:319-345: single_* spaces per env (Dict of Box per observation group, Box for the action) and their
        batched forms.  gymnasium is used when importable; otherwise plain stand-ins with the same ``shape / low / high / dtype``."""
        try:
            import gymnasium as gym

            box = lambda shape: gym.spaces.Box(low=-np.inf, high=np.inf, shape=shape)  # noqa: E731
            dct = gym.spaces.Dict
        except ImportError:
            import types

            box = lambda shape: types.SimpleNamespace(low=-np.inf, high=np.inf, shape=tuple(shape), dtype=np.float32)  # noqa: E731
            dct = dict
        N = self.num_envs
        om = self.observation_manager

        def spaces(batch: tuple):  # a Box per concatenated group (its group_obs_dim), a Dict of term Boxes otherwise
            out = {}
            for gname, tnames in om.active_terms.items():
                gdim = om.group_obs_dim[gname]
                if om.group_obs_concatenate[gname]:
                    out[gname] = box(batch + tuple(gdim))
                else:
                    out[gname] = dct({t: box(batch + tuple(d)) for t, d in zip(tnames, gdim)})
            return dct(out)

        self.single_observation_space = spaces(())
        self.single_action_space = box((self.plan.action_dim,))
        self.observation_space = spaces((N,))
        self.action_space = box((N, self.plan.action_dim))

    def close(self):
        if getattr(self, "_plan_h", None):
            self._lib.imx_plan_destroy(self._plan_h)
            self._plan_h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- internals -------------------------------------------------------------------------------------------------
    def _install_term_cfg(self, section: str, term_name: str, entry: dict):
        old = self._cfg_dict[section][term_name]
        new = dict(old)
        new.update({k: v for k, v in entry.items() if k in old or k in ("weight", "params", "func", "time_out")})
        if new == old:
            return
        if func_name_of(new.get("func")) != func_name_of(old.get("func")):
            raise NotImplementedError(f"set_term_cfg('{term_name}'): replacing a term's function needs a new environment")
        self._cfg_dict[section][term_name] = new
        try:
            plan = compile_plan(self._cfg_dict, self._robot)
            blob = np.ascontiguousarray(plan.blob, np.int32)
            check(self._lib.imx_plan_update(self._plan_h, blob.ctypes.data, blob.size, _lib.current_stream(self.device)))
        except Exception:
            self._cfg_dict[section][term_name] = old
            raise
        # the python-side mirror (term weights / params read by the Python-evaluated route and by get_term_cfg)
        for cur, fresh in zip(self.plan.reward_terms, plan.reward_terms):
            cur.weight, cur.params = fresh.weight, fresh.params
        for cur, fresh in zip(self.plan.termination_terms, plan.termination_terms):
            cur.params, cur.time_out = fresh.params, fresh.time_out
        self.plan.blob = plan.blob
        for kind in ("rew", "term"):
            for t, _ in self._ext_funcs[kind]:
                self._resolve_ext(t)  # call-time parameters of Python-evaluated terms follow the new cfg

    def _resolve_ext(self, term):
        """Callable + call-time parameters of a Python-evaluated term (ManagerBase._resolve_common_term_cfg,
        managers/manager_base.py:278-395: SceneEntityCfg parameters are resolved against the scene once)."""
        from .plan import PlanCompiler

        import inspect

        f = term.external
        fn = _string_to_callable(f) if isinstance(f, str) else f
        comp = PlanCompiler(self._cfg_dict, self._robot)
        term.call_params = {k: (_SceneEntityView(v, comp) if _looks_like_scene_entity(v) else v) for k, v in term.params.items()}
        term.py_mod_funcs = [(_string_to_callable(m) if isinstance(m, str) else m, mp) for m, mp in term.py_modifiers]
        if inspect.isclass(fn):
            # a class term (manager_base.py:324-327,393-395): instantiated ONCE with (cfg, env); the instance is what gets called every
            # step, and its reset(env_ids) runs for the envs of every reset (reward_manager.py:123-124, observation_manager.py:224)
            inst = getattr(term, "class_instance", None)
            if inst is None:
                entry = {"func": fn, "params": term.call_params}
                if hasattr(term, "weight"):
                    entry["weight"] = term.weight
                inst = fn(cfg=TermCfgView(entry), env=self)
                if not callable(inst):
                    raise TypeError(f"class term '{term.name}': {fn.__name__} instances are not callable (ManagerTermBase.__call__)")
                term.class_instance = inst
                self._class_terms.append(inst)
            return inst
        return fn

    def _state(self) -> ImxState:
        idx = self.feed.index
        st = self._state_cache.get(idx)
        if st is None:
            snap = self.feed.snapshot(idx)
            kw = {n: snap[n].data_ptr() for n in _lib.STATE_FIELDS if n in snap}
            if self.command_term is not None:
                kw["command"] = self.command_term.vel_command_b.data_ptr()
                kw["command_time_left"] = self.command_term.time_left.data_ptr()
                kw["command_counter"] = self.command_term.command_counter.data_ptr()
            if self.articulation is not None:  # ArticulationData's own buffers (fixed addresses), filled by imx_articulation_update
                ar = self.articulation
                kw.update(root_pos_w=ar.root_pos_w.data_ptr(), root_quat_w=ar.root_quat_w.data_ptr(), root_lin_vel_w=ar.root_lin_vel_w.data_ptr(),
                          root_ang_vel_w=ar.root_ang_vel_w.data_ptr(), joint_acc=ar.joint_acc.data_ptr())
            if self.terrain_importer is not None:  # scene.env_origins is the importer's tensor (the curriculum moves it)
                kw["env_origins"] = self.terrain_importer.env_origins.data_ptr()
            if self.contact_sensor is not None:
                d = self.contact_sensor.data
                kw.update(net_forces_w_history=d.net_forces_w_history.data_ptr(), last_air_time=d.last_air_time.data_ptr(),
                          current_air_time=d.current_air_time.data_ptr(), current_contact_time=d.current_contact_time.data_ptr())
            kw["ext_reward"] = _lib.ptr(self._ext_reward)
            kw["ext_term"] = _lib.ptr(self._ext_term)
            kw["ext_obs"] = _lib.ptr(self._ext_obs)
            st = ImxState(**kw)
            self._state_cache[idx] = st
        return st

    def _root_frame(self) -> dict:
        N, f = self.num_envs, self.feed
        out = {k: torch.empty(N, 3, device=self.device) for k in ("root_lin_vel_b", "root_ang_vel_b", "projected_gravity_b")}
        g = self.plan.gravity_dir
        check(self._lib.imx_root_frame(N, f["root_quat_w"].data_ptr(), f["root_lin_vel_w"].data_ptr(),
                                       f["root_ang_vel_w"].data_ptr(), g[0], g[1], g[2], out["root_lin_vel_b"].data_ptr(),
                                       out["root_ang_vel_b"].data_ptr(), out["projected_gravity_b"].data_ptr(),
                                       _lib.current_stream(self.device)))
        return out

    def _process_action(self, action: torch.Tensor):
        A = self.plan.action_dim
        if action.shape[1] != A:  # action_manager.py:328-329
            raise ValueError(f"Invalid action shape, expected: {A}, received: {action.shape[1]}.")
        a = action.to(self.device, torch.float32).contiguous()
        clip = math.inf if self.clip_actions is None else float(self.clip_actions)
        check(self._lib.imx_action_process(self._plan_h, self.num_envs, a.data_ptr(), clip, ctypes.byref(self._state()),
                                           ctypes.byref(self._bufs), _lib.current_stream(self.device)))

    def _eval_external(self, kind: str):
        for col, (term, fn) in enumerate(self._ext_funcs[kind] if kind != "obs" else ()):  # (each term is called ONCE per step: class terms keep state)
            if kind == "rew" and term.weight == 0.0:
                continue  # reward_manager.py:145: a zero-weight term is not evaluated
            val = fn(self, **term.call_params)
            if kind == "rew":
                self._ext_reward[:, col] = val
            elif kind == "term":
                self._ext_term[:, col] = val
        if kind == "obs":
            c = 0
            for term, fn in self._ext_funcs["obs"]:
                val = fn(self, **term.call_params)
                for mfn, mp in term.py_mod_funcs:  # a foreign (function-style) modifier chain: observation_manager.py:310-312
                    val = mfn(val, **mp)
                self._ext_obs[:, c:c + term.dim] = val.reshape(self.num_envs, -1)
                c += term.dim

    def _compute_observations(self, fill_history: bool = False, frame_current: bool = False, finish_step_tail: bool = False) -> torch.Tensor:
        if self._ext_funcs["obs"]:
            self._eval_external("obs")
        hits = None
        if self.materialize_ray_hits and self.plan.num_rays:
            if self._ray_hits is None:
                self._ray_hits = torch.empty(self.num_envs, self.plan.num_rays, 3, device=self.device)
            hits = self._ray_hits.data_ptr()
        self._bufs.scan_drift_feed = _lib.ptr(self._scan_drift_feed)
        check(self._lib.imx_observations(
            self._plan_h, self.num_envs, ctypes.byref(self._state()), ctypes.byref(self._bufs),
            self.terrain.handle if self.terrain is not None else None, _lib.ptr(self._noise_u), self.noise_seed,
            (1 if self.plan.enable_corruption else 0) | (2 if fill_history else 0) | (4 if frame_current else 0)
            | (8 if self.scanner_keep_all_hits else 0) | (16 if finish_step_tail else 0), hits, _lib.current_stream(self.device)))
        return self._obs

    @property
    def _has_orchestration(self) -> bool:
        return self.event_manager is not None or self.command_term is not None or self.curriculum_manager is not None

    def attach_actuator(self, actuator_net):
        """An env-owned ``producers.ActuatorNetLSTM``: its hidden / cell state restarts with the env (scene.reset -> Articulation.reset ->
        ActuatorNetLSTM.reset, actuators/actuator_net.py:66-70) inside the orchestration launch."""
        self.actuator_net = actuator_net
        self._orch = None

    def _build_orch(self):
        from ._lib import ImxOrch

        f, p = self.feed, _lib.ptr
        o = ImxOrch(num_envs=self.num_envs, num_joints=self.plan.num_joints, num_bodies=self.plan.robot.num_bodies,
                    step_counter_d=self._counters[2:3].data_ptr(), dt=float(self.step_dt),
                    default_root_state_d=p(self.default_root_state), default_joint_pos_d=p(f["default_joint_pos"]),
                    default_joint_vel_d=p(f["default_joint_vel"]), soft_joint_pos_limits_d=p(f["soft_joint_pos_limits"]),
                    soft_joint_vel_limits_d=p(f["soft_joint_vel_limits"]),
                    env_origins_d=p(self.terrain_importer.env_origins if self.terrain_importer is not None else f["env_origins"]),
                    root_pose_out_d=p(self.sim_writes["root_pose"]), root_vel_out_d=p(self.sim_writes["root_vel"]),
                    joint_pos_out_d=p(self.sim_writes["joint_pos"]), joint_vel_out_d=p(self.sim_writes["joint_vel"]),
                    ext_force_out_d=p(self.sim_writes["ext_force"]), ext_torque_out_d=p(self.sim_writes["ext_torque"]),
                    ev_part_d=p(self._ev_part))
        if self.event_manager is not None:
            self.event_manager.fill(o)
        if self.curriculum_manager is not None and self.curriculum_manager.active_terms:
            ti = self.terrain_importer
            o.terrain_origins_d, o.terrain_types_d, o.terrain_levels_d = p(ti.terrain_origins), p(ti.terrain_types), p(ti.terrain_levels)
            o.terrain_rows, o.terrain_cols = int(ti.terrain_origins.shape[0]), int(ti.terrain_origins.shape[1])
            o.terrain_size_x, o.max_episode_length_s = float(ti.size_x), float(self.max_episode_length_s)
        ct = self.command_term
        if ct is not None:
            o.has_command, o.heading_command = 1, int(ct.heading_command)
            for k, v in enumerate(ct._cfg15):
                o.command_cfg[k] = float(v)
            o.vel_command_b_d, o.heading_target_d = p(ct.vel_command_b), p(ct.heading_target)
            o.is_heading_env_d, o.is_standing_env_d = p(ct.is_heading_env), p(ct.is_standing_env)
            o.command_time_left_d, o.command_counter_d = p(ct.time_left), p(ct.command_counter)
            o.metric_error_vel_xy_d, o.metric_error_vel_yaw_d = p(ct.metrics["error_vel_xy"]), p(ct.metrics["error_vel_yaw"])
        cs = self.contact_sensor
        if cs is not None:
            d = cs.data
            o.cs_timestamp_d, o.cs_timestamp_last_update_d, o.cs_is_outdated_d = p(cs._timestamp), p(cs._timestamp_last_update), p(cs._is_outdated)
            o.cs_net_forces_w_d = p(d.net_forces_w)
            o.cs_net_forces_w_history_d = p(d.net_forces_w_history) if cs.history_length > 0 else None
            if cs.cfg.track_air_time:
                o.cs_last_air_time_d, o.cs_current_air_time_d = p(d.last_air_time), p(d.current_air_time)
                o.cs_last_contact_time_d, o.cs_current_contact_time_d = p(d.last_contact_time), p(d.current_contact_time)
            o.cs_num_bodies, o.cs_history_length = cs.num_bodies, cs.history_length
        an = self.actuator_net
        if an is not None:
            o.lstm_hidden_d, o.lstm_cell_d = p(an.sea_hidden_state), p(an.sea_cell_state)
            o.lstm_layers, o.lstm_hidden_dim = an.num_layers, an.hidden_dim
        return o

    def _orchestrate(self, reset_mask, do_step: bool):
        """``_reset_idx`` for the flagged envs (+, inside step(), CommandManager.compute and the interval events): ``imx_reset_orchestrate``."""
        if self._orch is None:
            self._orch = self._build_orch()
        o, f = self._orch, self.feed
        o.reset_mask_d = _lib.ptr(reset_mask)
        o.do_step = 1 if do_step else 0
        o.seed = self.noise_seed
        src = self.articulation if self.articulation is not None else None
        get = (lambda n: getattr(src, n)) if src is not None else (lambda n: f[n])
        o.root_pos_w_d, o.root_quat_w_d = get("root_pos_w").data_ptr(), get("root_quat_w").data_ptr()
        o.root_lin_vel_w_d, o.root_ang_vel_w_d = get("root_lin_vel_w").data_ptr(), get("root_ang_vel_w").data_ptr()
        if self.event_manager is not None:
            for i, t in enumerate(self.event_manager.terms):  # (parity runs re-point these between steps)
                o.terms[i].uniforms_d, o.terms[i].interval_uniforms_d = _lib.ptr(t.uniforms), _lib.ptr(t.interval_uniforms)
        o.command_uniforms_d = _lib.ptr(self._orch_draws.get("command"))
        o.rand_levels_d = _lib.ptr(self._orch_draws.get("rand_levels"))
        check(self._lib.imx_reset_orchestrate(ctypes.byref(o), _lib.current_stream(self.device)))

    # ---- MDP operations ------------------------------------------------------------------------------------------
    def reset(self, seed: int | None = None, env_ids: Sequence[int] | None = None, options: dict | None = None):
        """ManagerBasedEnv.reset (manager_based_env.py:264-315): reset every env, return (obs_dict, extras)."""
        if seed is not None:
            self.seed(seed)
        ids = slice(None) if env_ids is None else env_ids
        log = {}
        log.update(self.reward_manager.reset(ids))
        log.update(self.termination_manager.reset(ids))  # (_reset_idx collects every manager's entries, manager_based_rl_env.py:365-389)
        self.action_manager.reset(ids)
        self._episode_length_buf[ids] = 0
        self.extras["log"] = log
        for inst in self._class_terms:  # every manager's reset(env_ids) reaches its class terms (manager_based_env.py:264-315 -> _reset_idx)
            if hasattr(inst, "reset"):
                inst.reset(env_ids=None if env_ids is None else env_ids)
        f = self.feed
        if self.articulation is not None:  # ArticulationData of the state the env starts from
            self.articulation.update(f.physx("root_transforms"), f.physx("root_velocities"), f["joint_vel"], self.step_dt)
        if self.contact_sensor is not None:
            self.contact_sensor.reset(None if env_ids is None else env_ids)
        if self.actuator_net is not None:
            self.actuator_net.reset(None if env_ids is None else env_ids)
        if self._has_orchestration:
            # _reset_idx(env_ids) (manager_based_rl_env.py:347-392): curriculum, scene.reset, reset events, manager resets with their
            # log entries -- no command compute, no interval events (those belong to step()).  Host-side reductions are fine here
            mask = None
            if env_ids is not None:
                mask = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
                mask[ids] = True
            if self.command_term is not None:  # CommandTerm.reset logs the metrics before zeroing them (command_manager.py:123-149)
                cname = [k for k in self._log_index if k.startswith("Metrics/")][0].split("/")[1]
                for m in ("error_vel_xy", "error_vel_yaw"):
                    log[f"Metrics/{cname}/{m}"] = torch.mean(self.command_term.metrics[m][ids])
            self._orchestrate(mask, do_step=False)
            if self.curriculum_manager is not None:
                log.update(self.curriculum_manager.reset(ids))
            for k, v in log.items():  # the device-side log entries follow (the step tail only refreshes them on steps with resets)
                if k in self._log_index:
                    self._log_out[self._log_index[k]] = v
        # ObservationManager.reset -> CircularBuffer.reset: the history windows of the reset envs restart from this observation
        if env_ids is None:
            self._compute_observations(fill_history=True)
        else:
            self.reset_buf.zero_()
            self.reset_buf[ids] = True
            self._compute_observations()
        return self.observation_manager.shaped(), self.extras

    def step(self, action: torch.Tensor):
        """ManagerBasedRLEnv.step (manager_based_rl_env.py:153-242)."""
        # -- pre-physics
        self._process_action(action)
        return self._step_after_action()

    def _step_after_action(self, rollout_slot=None):
        """``step()`` from the physics on.  The fused rollout (rsl_rl/runner.py) enters here: its actor kernel has already run the
        sampled action through the action terms (imx_mlp_infer_act), and ``rollout_slot`` (an ``ImxRolloutSlot``) makes the step kernel
        write slot t of the RolloutStorage itself (imx_terminations_rewards_rollout)."""
        # -- physics (decimation x sim.step) is replaced by the feed moving to its next recorded state.  What the reference runs around
        #    every physics step on the torch side stays: the actuator model before it (scene.write_data_to_sim -> Articulation.
        #    _apply_actuator_model -> ActuatorNetLSTM.compute, manager_based_rl_env.py:182-196) ...
        if self.actuator_net is not None:
            f = self.feed
            for _ in range(int(self.cfg_decimation)):
                self.actuator_net.compute(self._processed_action, f["joint_pos"], f["joint_vel"])
        self._sim_step_counter += int(self.cfg_decimation)
        self.feed.advance()
        if self.articulation is not None:
            # ... and scene.update(dt) after it: ArticulationData refreshed from the PhysX-layout tensors (root transforms XYZW, root
            # velocities, dof velocities): imx_articulation_update; the step kernels read ITS root state and joint_acc
            f = self.feed
            self.articulation.update(f.physx("root_transforms"), f.physx("root_velocities"), f["joint_vel"], self.step_dt)
        if self.contact_sensor is not None:
            # scene.update -> ContactSensor.update: the feed delivers one force sample per env step (history slot 0)
            self.contact_sensor.update(self.feed["net_forces_w_history"][:, 0], self.step_dt)
        self.common_step_counter += 1
        # -- post-physics: counters, terminations, rewards, reset bookkeeping (one kernel)
        if self._ext_funcs["rew"] or self._ext_funcs["term"]:
            self._eval_external("term")
            self._eval_external("rew")
        if self._has_orchestration and not self.defer_step_tail:
            raise RuntimeError("an env that owns its Event / Command / Curriculum managers needs the deferred step tail (defer_step_tail=True): "
                               "the Metrics/* and Curriculum/* log entries are reduced after the orchestration launch")
        # flag 1: the end of the step (ordered reset ids, reset count, Episode_* log) is finished by an extra workgroup of the
        # observation kernel below -- same stream, kernel boundary in between -- instead of a fence + ticket in this launch
        self._bufs.scan_drift_feed = _lib.ptr(self._scan_drift_feed)  # the step kernel advances the sensor clock (resets draw a drift)
        check(self._lib.imx_terminations_rewards_rollout(self._plan_h, self.num_envs, ctypes.byref(self._state()), ctypes.byref(self._bufs),
                                                         1 if self.defer_step_tail else 0,
                                                         ctypes.byref(rollout_slot) if rollout_slot is not None else None,
                                                         _lib.current_stream(self.device)))
        self.extras["log"] = self._log_views
        if self._class_terms:
            # _reset_idx reaches the class terms between the reward and the observation pass (manager_based_rl_env.py:215-218 ->
            # RewardManager.reset / ObservationManager.reset).  This route is Python-evaluated anyway: the ids are read like the
            # reference reads them (reset_buf.nonzero: one host round trip)
            rids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
            if len(rids) > 0:
                for inst in self._class_terms:
                    if hasattr(inst, "reset"):
                        inst.reset(env_ids=rids)
        # -- _reset_idx for the reset envs (curriculum, scene.reset, reset events, command reset + Metrics / Curriculum log sums), then
        #    CommandManager.compute(dt) and the interval events: one launch, or commands from the feed
        if self._has_orchestration:
            self._orchestrate(self.reset_buf, do_step=True)
        # -- observations on the post-reset state (one kernel, ray-cast fused); imx_terminations_rewards left the frame table of
        #    this state snapshot behind (the feed's root state is not rewritten by the reset events: they go to sim_writes)
        self._compute_observations(frame_current=True, finish_step_tail=self.defer_step_tail)
        return self.observation_manager.shaped(), self._reward_buf, self.reset_terminated, self.reset_time_outs, self.extras

    @property
    def cfg_decimation(self) -> int:
        c = self.cfg
        if isinstance(c, dict):
            c = c.get("env", c)
            return int(c.get("decimation", 1))
        return int(getattr(c, "decimation", 1))

    @property
    def is_finite_horizon(self) -> bool:
        return self.plan.is_finite_horizon
