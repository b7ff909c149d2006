"""TEST INFRASTRUCTURE (build container only): generate ``tests/golden/*.npz`` + ``isaaclab_amd/configs/*.json``
by running the REAL reference managers / mdp terms / task cfgs (imported from /root/reference through
``oracle/ref_import.py``) on the seeded synthetic state feed.

    python oracle/gen_golden.py            # regenerates every fixture

What runs from the reference (unmodified, imported, not copied):
  * task cfgs: CartpoleEnvCfg, AnymalCFlatEnvCfg, AnymalCRoughEnvCfg, G1RoughEnvCfg  -> ``cfg.to_dict()`` fixtures
  * ActionManager/JointAction, TerminationManager, RewardManager, ObservationManager (+ uniform_noise)
  * every mdp term the cfgs name, ``ArticulationData`` derived properties (root_lin_vel_b, projected_gravity_b ...),
    ``ContactSensor.compute_first_contact``, ``grid_pattern``, ``quat_apply_yaw``, ``convert_height_field_to_mesh``,
    and ``isaaclab.utils.math`` helpers on random + edge inputs.
What cannot run (absent third-party): Warp ``mesh_query_ray`` -> ray hits come from the fp64
brute-force oracle (``oracle/raycast_oracle.c``); rsl_rl -> no GAE/PPO goldens (parity unpinned).

The step emulation follows ``ManagerBasedRLEnv.step`` (isaaclab/envs/manager_based_rl_env.py:153-242) and
``_reset_idx`` (:347-392) with PhysX/scene/events/commands replaced by the feed.
"""

from __future__ import annotations

import hashlib
import json
import math
import os
import re
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402

ref_import.install()

import isaaclab.utils.math as ref_math  # noqa: E402
import isaaclab.utils.noise.noise_model as ref_noise_model  # noqa: E402
import isaaclab.utils.string as ref_string  # noqa: E402
from isaaclab.assets.articulation.articulation_data import ArticulationData  # noqa: E402
from isaaclab.managers import ActionManager, ObservationManager, RewardManager, TerminationManager  # noqa: E402
from isaaclab.sensors.contact_sensor import ContactSensor  # noqa: E402
from isaaclab.sensors.ray_caster.patterns import patterns as ref_patterns  # noqa: E402
from isaaclab.terrains.height_field.utils import convert_height_field_to_mesh  # noqa: E402

from isaaclab_amd.robots import ANYMAL_C, CARTPOLE, G1, RobotSpec  # noqa: E402
from isaaclab_amd.state_feed import DYNAMIC, EXTRA, STATIC, StateFeed  # noqa: E402
from isaaclab_amd.terrain import make_rough_terrain  # noqa: E402
from oracle.raycast import raycast_f64  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CONFIGS = os.path.join(ROOT, "isaaclab_amd", "configs")


# --------------------------------------------------------------------------- cfg -> JSON
def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, slice):
        return f"slice({x.start}, {x.stop}, {x.step})"
    if isinstance(x, str):  # a path built from a mocked carb setting carries the mock's id(): scrub it (reproducible JSON)
        return re.sub(r"<_MockModule name='([^']*)' id='\d+'>", r"<mock:\1>", x)
    if isinstance(x, (int, bool)) or x is None:
        return x
    if isinstance(x, float):
        if math.isinf(x) or math.isnan(x):
            return str(x)
        return x
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, torch.Tensor):
        return x.tolist()
    return _jsonable(str(x))


def dump_cfg(task: str, env_cfg, agent_cfg, robot: RobotSpec):
    d = env_cfg.to_dict()
    keep = {k: d[k] for k in ("decimation", "episode_length_s", "is_finite_horizon", "observations", "actions",
                              "rewards", "terminations", "commands", "seed") if k in d}
    keep["sim"] = {"dt": d["sim"]["dt"], "gravity": d["sim"].get("gravity", (0.0, 0.0, -9.81))}
    scene = d["scene"]
    keep["scene"] = {"num_envs": scene["num_envs"], "env_spacing": scene["env_spacing"]}
    for name in ("height_scanner", "contact_forces"):
        if scene.get(name) is not None:
            s = dict(scene[name])
            s.pop("visualizer_cfg", None)
            keep["scene"][name] = s
    if scene.get("terrain") is not None:
        t = scene["terrain"]
        tg = t.get("terrain_generator")
        keep["scene"]["terrain"] = {
            "terrain_type": t.get("terrain_type"),
            "terrain_generator": None if tg is None else {k: tg[k] for k in ("size", "border_width", "num_rows",
                                                                            "num_cols", "horizontal_scale",
                                                                            "vertical_scale", "slope_threshold")},
        }
    out = {"task": task, "robot": robot.name, "env": keep, "agent": agent_cfg.to_dict()}
    os.makedirs(CONFIGS, exist_ok=True)
    with open(os.path.join(CONFIGS, task + ".json"), "w") as f:
        json.dump(_jsonable(out), f, indent=1, sort_keys=False)
    return out


# --------------------------------------------------------------------------- fake scene
class FakeArticulationData:
    """Plain tensors for the PhysX-sourced buffers; DERIVED quantities run the reference's own property code."""

    root_lin_vel_b = ArticulationData.root_lin_vel_b
    root_ang_vel_b = ArticulationData.root_ang_vel_b
    projected_gravity_b = ArticulationData.projected_gravity_b

    def __init__(self, feed: StateFeed):
        self._feed = feed
        g = torch.tensor(feed.gravity_dir, dtype=torch.float32)
        self.GRAVITY_VEC_W = ref_math.normalize(g.unsqueeze(0)).squeeze(0).repeat(feed.num_envs, 1)

    def __getattr__(self, name):
        if name == "root_link_quat_w":
            name = "root_quat_w"
        try:
            return self._feed[name]
        except KeyError:
            raise AttributeError(name)


class FakeArticulation:
    def __init__(self, robot: RobotSpec, feed: StateFeed):
        self.robot = robot
        self.data = FakeArticulationData(feed)
        self.joint_names = list(robot.joint_names)
        self.body_names = list(robot.body_names)
        self.num_joints = robot.num_joints
        self.num_bodies = robot.num_bodies
        self.num_instances = feed.num_envs
        self.device = "cpu"
        self.targets = {}

    def find_joints(self, name_keys, joint_subset=None, preserve_order=False):
        return ref_string.resolve_matching_names(name_keys, self.joint_names, preserve_order)

    def find_bodies(self, name_keys, preserve_order=False):
        return ref_string.resolve_matching_names(name_keys, self.body_names, preserve_order)

    def set_joint_position_target(self, target, joint_ids=None):
        self.targets["pos"] = target

    def set_joint_effort_target(self, target, joint_ids=None):
        self.targets["effort"] = target


class FakeContactSensor:
    compute_first_contact = ContactSensor.compute_first_contact

    def __init__(self, robot: RobotSpec, feed: StateFeed):
        self.cfg = types.SimpleNamespace(track_air_time=True)
        self.body_names = list(robot.body_names)
        self.num_bodies = robot.num_bodies
        self._feed = feed
        outer = self

        class _Data:
            def __getattr__(self, name):
                return outer._feed[name]

        self.data = _Data()

    def find_bodies(self, name_keys, preserve_order=False):
        return ref_string.resolve_matching_names(name_keys, self.body_names, preserve_order)


class FakeRayCaster:
    def __init__(self):
        self.data = types.SimpleNamespace(pos_w=None, quat_w=None, ray_hits_w=None)


def make_real_ray_caster(scanner_cfg, feed: StateFeed, mesh):
    """The REAL ``RayCaster`` (``reset``, ``_initialize_rays_impl``, ``_update_buffers_impl``, lazy ``data``) and ``SensorBase``
    (``update``, ``_update_outdated_buffers``) over a stand-in prim view that serves the feed's root pose; only the Warp query
    ``raycast_mesh`` is replaced (fp64 brute force, oracle/raycast_oracle.c).  Built without ``__init__`` (which needs the simulator)."""
    import importlib

    rc_mod = importlib.import_module("isaaclab.sensors.ray_caster.ray_caster")  # (the package re-exports shadow the attribute chain)
    RayCasterData = importlib.import_module("isaaclab.sensors.ray_caster.ray_caster_data").RayCasterData

    class _View:  # plays isaacsim XFormPrim: get_world_poses(env_ids) -> (pos, quat wxyz)
        count = feed.num_envs

        def get_world_poses(self, env_ids):
            return feed["root_pos_w"][env_ids], feed["root_quat_w"][env_ids]

    rc_mod.XFormPrim = _View

    def fake_raycast_mesh(ray_starts, ray_directions, mesh, max_dist=1e6, **kw):
        verts, tris = mesh
        shape = ray_starts.shape
        hits, _, _ = raycast_f64(verts, tris, ray_starts.reshape(-1, 3).numpy(), ray_directions.reshape(-1, 3).numpy(), max_dist=max_dist)
        return torch.from_numpy(hits).view(shape), None, None, None

    rc_mod.raycast_mesh = fake_raycast_mesh
    N = feed.num_envs
    s = object.__new__(rc_mod.RayCaster)
    s.cfg = scanner_cfg
    s._view = _View()
    s._device = "cpu"
    s._num_envs = N
    s._is_visualizing = False
    s._data = RayCasterData()
    s.meshes = {scanner_cfg.mesh_prim_paths[0]: mesh}
    s._timestamp = torch.zeros(N)  # SensorBase._initialize_impl (sensor_base.py)
    s._timestamp_last_update = torch.zeros(N)
    s._is_outdated = torch.ones(N, dtype=torch.bool)
    s._initialize_rays_impl()
    return s


class FakeScene:
    def __init__(self, entities: dict, sensors: dict, env_origins, cfg):
        self._e = dict(entities)
        self._e.update(sensors)
        self.sensors = sensors
        self.env_origins = env_origins
        self.cfg = cfg
        self.articulations = entities

    def keys(self):
        return list(self._e.keys())

    def __getitem__(self, k):
        return self._e[k]


class FakeCommandManager:
    def __init__(self, feed):
        self._feed = feed

    def get_command(self, name):
        return self._feed["command"]

    def get_term(self, name):  # what command_resample reads (terminations.py:41): CommandTerm.time_left / command_counter
        return types.SimpleNamespace(time_left=self._feed["command_time_left"], command_counter=self._feed["command_counter"])


def build_ref_env(env_cfg, robot: RobotSpec, feed: StateFeed, real_scanner=None):
    N = feed.num_envs
    robot_asset = FakeArticulation(robot, feed)
    sensors = {}
    if getattr(env_cfg.scene, "contact_forces", None) is not None:
        sensors["contact_forces"] = FakeContactSensor(robot, feed)
    if getattr(env_cfg.scene, "height_scanner", None) is not None:
        sensors["height_scanner"] = FakeRayCaster()
        pc = env_cfg.scene.height_scanner.pattern_cfg
        R = len(pc.func(pc, "cpu")[1])
        sensors["height_scanner"].data.pos_w = torch.zeros(N, 3)
        sensors["height_scanner"].data.ray_hits_w = torch.zeros(N, R, 3)
        if real_scanner is not None:
            sensors["height_scanner"] = real_scanner
    env = types.SimpleNamespace()
    env.num_envs = N
    env.device = "cpu"
    env.sim = types.SimpleNamespace(is_playing=lambda: True)
    env.cfg = env_cfg
    env.scene = FakeScene({"robot": robot_asset}, sensors, feed["env_origins"], env_cfg.scene)
    env.scene.terrain = types.SimpleNamespace(cfg=getattr(env_cfg.scene, "terrain", None))  # terrain_out_of_bounds reads scene.terrain.cfg.terrain_generator
    env.common_step_counter = 0
    env.step_dt = env_cfg.sim.dt * env_cfg.decimation
    env.max_episode_length_s = env_cfg.episode_length_s
    env.max_episode_length = math.ceil(env_cfg.episode_length_s / env.step_dt)
    env.episode_length_buf = torch.zeros(N, dtype=torch.long)
    env.command_manager = FakeCommandManager(feed)
    env.action_manager = ActionManager(env_cfg.actions, env)
    env.termination_manager = TerminationManager(env_cfg.terminations, env)
    env.reward_manager = RewardManager(env_cfg.rewards, env)
    env.observation_manager = ObservationManager(env_cfg.observations, env)
    return env


# --------------------------------------------------------------------------- ray caster (reference pieces + oracle hits)
def ref_height_scanner(scanner_cfg, feed: StateFeed, mesh):
    """ray_caster.py:201-260 with the Warp query replaced by the fp64 brute-force oracle."""
    starts, dirs = scanner_cfg.pattern_cfg.func(scanner_cfg.pattern_cfg, "cpu")
    R = len(dirs)
    offset_pos = torch.tensor(list(scanner_cfg.offset.pos))
    offset_quat = torch.tensor(list(scanner_cfg.offset.rot))
    dirs = ref_math.quat_apply(offset_quat.repeat(R, 1), dirs)
    starts = starts + offset_pos
    N = feed.num_envs
    starts = starts.repeat(N, 1, 1)
    dirs = dirs.repeat(N, 1, 1)
    pos_w = feed["root_pos_w"].clone()
    quat_w = feed["root_quat_w"].clone()
    assert scanner_cfg.attach_yaw_only
    ray_starts_w = ref_math.quat_apply_yaw(quat_w.repeat(1, R), starts)
    ray_starts_w += pos_w.unsqueeze(1)
    verts, tris = mesh
    hits, _, _ = raycast_f64(verts, tris, ray_starts_w.numpy().reshape(-1, 3), dirs.numpy().reshape(-1, 3),
                             max_dist=scanner_cfg.max_distance)
    return pos_w, quat_w, ray_starts_w, dirs, torch.from_numpy(hits).view(N, R, 3)


# --------------------------------------------------------------------------- one task
def mesh_sha256(vertices, triangles) -> str:
    """Hash of the terrain mesh a fixture was generated on (tests/test_oracle_golden.py re-derives it from the terrain
    generator at HEAD, so that a change of ``make_rough_terrain`` cannot strand a committed fixture unnoticed)."""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(vertices, np.float32).tobytes())
    h.update(np.ascontiguousarray(triangles, np.uint32).tobytes())
    return h.hexdigest()


ROUGH_TERRAIN_ARGS = dict(num_rows=2, num_cols=3, tile=4.0, border=3.0, seed=11)  # the small terrain of the rough fixtures


def run_task(task: str, env_cfg, agent_cfg, robot: RobotSpec, N: int, steps: int, seed: int, mesh=None, extent=None, kitchen=None):
    """``kitchen``: options of the kitchen-sink fixture -- ``real_scanner`` (the height scanner is the real RayCaster/SensorBase with its
    update-period gating and drift), ``scan_ts0`` (timestamps injected after reset()), ``feed_tweak(feed)``, ``weight_change`` =
    (step, term, weight) applied with the real ``modify_reward_weight`` curriculum term before that step."""
    kitchen = kitchen or {}
    torch.manual_seed(seed)
    dump_cfg(task, env_cfg, agent_cfg, robot)
    feed = StateFeed(robot, N, "cpu", seed=seed, num_snapshots=steps + 1, extent_xy=extent)
    if kitchen.get("feed_tweak"):
        kitchen["feed_tweak"](feed)
    scanner = make_real_ray_caster(env_cfg.scene.height_scanner, feed, mesh) if kitchen.get("real_scanner") else None
    env = build_ref_env(env_cfg, robot, feed, real_scanner=scanner)
    has_scan = "height_scanner" in env.scene.sensors
    A = env.action_manager.total_action_dim
    om = env.observation_manager
    group_names = list(om.group_obs_dim)

    def flat_width(g):  # columns of the group in the fused row: every term flattened (history windows oldest first), side by side
        return int(sum(int(np.prod(d)) for d in om.group_obs_term_dim[g]))

    D0 = flat_width(group_names[0])
    D = sum(flat_width(g) for g in group_names)
    gen = torch.Generator().manual_seed(seed + 1000)
    rec: dict[str, np.ndarray] = {}

    def put(name, t):
        rec[name] = t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t)

    meta = dict(task=task, robot=robot.name, num_envs=N, steps=steps, seed=seed, action_dim=int(A), obs_dim=int(D0), obs_dim_total=int(D),
                obs_groups=group_names, obs_group_dims=[flat_width(g) for g in group_names],
                obs_group_shapes={g: (_jsonable(om.group_obs_dim[g])) for g in group_names},
                obs_group_term_shapes={g: [[int(x) for x in d] for d in om.group_obs_term_dim[g]] for g in group_names},
                obs_group_terms={g: list(om.active_terms[g]) for g in group_names},
                obs_group_concatenate={g: bool(om.group_obs_concatenate[g]) for g in group_names},
                step_dt=env.step_dt, max_episode_length=env.max_episode_length,
                max_episode_length_s=env.max_episode_length_s, gravity_dir=feed.gravity_dir,
                reward_terms=env.reward_manager.active_terms, termination_terms=env.termination_manager.active_terms,
                obs_terms=env.observation_manager.active_terms[group_names[0]],
                obs_term_dims=[[int(x) for x in d] for d in env.observation_manager.group_obs_term_dim[group_names[0]]])
    for n in STATIC:
        put("static/" + n, feed[n])
    in_names = DYNAMIC + (EXTRA if kitchen else ())
    if has_scan:
        put("mesh/vertices", mesh[0])
        put("mesh/triangles", mesh[1])
        meta["mesh_sha256"] = mesh_sha256(mesh[0], mesh[1])
        meta["terrain_args"] = ROUGH_TERRAIN_ARGS

    # uniform samples consumed by uniform_noise (observation_manager.py:313 -> noise_model.py:62): recorded so the
    # HIP path can be fed the same draws (torch's CPU RNG stream cannot be reproduced in-kernel).
    real_rand_like = torch.rand_like
    real_randn_like = torch.randn_like

    def compute_obs(tag):
        # the reference computes obs term by term; noisy terms call rand_like in term order.  Columns of `u` are
        # laid out per OBS COLUMN (D wide): advance `col` to each term's offset.
        u = torch.rand(N, D, generator=gen)
        # term-offset bookkeeping: patch so that draw k lands on the columns of the k-th noisy term (groups side by side, in the
        # order ObservationManager.compute walks them)
        noisy_offsets, base = [], 0
        for gname in group_names:
            dims = [int(np.prod(d)) for d in om.group_obs_term_dim[gname]]
            cfgs = om._group_obs_term_cfgs[gname]
            offs = np.concatenate([[0], np.cumsum(dims)])
            # (constant_noise draws nothing; gaussian_noise draws with randn_like: its columns of the recorded array hold normal samples)
            drawing = [(i, c) for i, c in enumerate(cfgs) if c.noise and c.noise.func.__name__ != "constant_noise"]
            noisy_offsets += [base + int(offs[i]) for i, c in drawing]
            for i, c in drawing:
                if c.noise.func.__name__ == "gaussian_noise":
                    u[:, base + int(offs[i]):base + int(offs[i]) + dims[i]] = torch.randn(N, dims[i], generator=gen)
            base += int(offs[-1])
        put(f"{tag}/noise_u", u)
        it = iter(noisy_offsets)

        def rand_like_at(x, *a, **k):
            c0 = next(it)
            return u[:, c0:c0 + x.shape[1]].clone()

        ref_noise_model.torch.rand_like = rand_like_at  # same module object as torch; restore below
        ref_noise_model.torch.randn_like = rand_like_at
        try:
            all_obs = env.observation_manager.compute()
        finally:
            torch.rand_like = real_rand_like
            torch.randn_like = real_randn_like
        obs = all_obs[group_names[0]]
        put(f"{tag}/obs", obs)
        for gname in group_names[1:]:
            if isinstance(all_obs[gname], dict):  # concatenate_terms = False: a dict of term tensors
                for tname, tv in all_obs[gname].items():
                    put(f"{tag}/obs/{gname}/{tname}", tv)
            else:
                put(f"{tag}/obs/{gname}", all_obs[gname])
        if scanner is not None:  # what the lazy refresh inside compute() left in the sensor
            put(f"{tag}/sensor_pos_w", scanner._data.pos_w)
            put(f"{tag}/ray_hits_w", scanner._data.ray_hits_w)
            put(f"{tag}/scan_timestamp", scanner._timestamp)
            put(f"{tag}/scan_timestamp_last_update", scanner._timestamp_last_update)
        return obs

    def update_scanner():
        if not has_scan:
            return
        if scanner is not None:
            # the hits every env WOULD get at its current pose + drift (input of the oracle's own gating; the real sensor decides
            # by itself, lazily, inside observation_manager.compute())
            R = scanner.num_rays
            starts_w = ref_math.quat_apply_yaw(feed["root_quat_w"].repeat(1, R), scanner.ray_starts) + (feed["root_pos_w"] + scanner.drift).unsqueeze(1)
            fresh, _, _ = raycast_f64(mesh[0], mesh[1], starts_w.numpy().reshape(-1, 3), scanner.ray_directions.numpy().reshape(-1, 3),
                                      max_dist=env_cfg.scene.height_scanner.max_distance)
            return starts_w, torch.from_numpy(fresh).view(N, R, 3)
        pos_w, quat_w, starts_w, dirs_w, hits = ref_height_scanner(env_cfg.scene.height_scanner, feed, mesh)
        s = env.scene.sensors["height_scanner"]
        s.data.pos_w, s.data.quat_w, s.data.ray_hits_w = pos_w, quat_w, hits
        return starts_w, hits

    # ---- reset() : ManagerBasedEnv.reset (manager_based_env.py:264-315) -> obs only
    for n in in_names:
        put(f"reset/in/{n}", feed[n])
    env.action_manager.reset(torch.arange(N))  # ManagerBasedEnv.reset -> _reset_idx(all) -> ActionManager.reset (an EMA term starts from the joint positions)
    if scanner is not None:  # ManagerBasedEnv.reset -> _reset_idx -> scene.reset(env_ids) -> RayCaster.reset (drift drawn here)
        scanner.reset(torch.arange(N))
        put("reset/scan_drift", scanner.drift)
    sc = update_scanner()
    if sc is not None:
        put("reset/ray_starts_w", sc[0])
        put("reset/ray_hits_fresh" if scanner is not None else "reset/ray_hits_w", sc[1])
    compute_obs("reset")
    if kitchen.get("scan_ts0") is not None:  # long-running sensors: fp32 timestamps far from zero (the state a 16 s old episode has)
        scanner._timestamp[:] = kitchen["scan_ts0"]
        scanner._timestamp_last_update[:] = kitchen["scan_ts0"]
        put("reset/scan_ts0", scanner._timestamp)
    # random episode lengths (RSL-RL init_at_random_ep_len) incl. some that time out on step 1..steps
    ep = torch.randint(0, env.max_episode_length, (N,), generator=gen)
    ep[::17] = env.max_episode_length - 1
    ep[5::23] = env.max_episode_length - 2
    env.episode_length_buf[:] = ep
    put("reset/episode_length_buf", env.episode_length_buf)

    for t in range(steps):
        tag = f"step{t}"
        action = (torch.randn(N, A, generator=gen)).clamp(-3, 3)
        put(f"{tag}/action", action)
        for wc in kitchen.get("weight_changes", ()):
            if wc[0] == t:  # CurriculumManager term (envs/mdp/curriculums.py:20-37), the REAL function
                from isaaclab.envs.mdp.curriculums import modify_reward_weight

                env.common_step_counter = 10
                modify_reward_weight(env, torch.arange(N), term_name=wc[1], weight=wc[2], num_steps=5)
        # -- pre-physics
        env.action_manager.process_action(action)
        put(f"{tag}/processed_actions", torch.cat([t_.processed_actions for t_ in env.action_manager._terms.values()], dim=1))
        put(f"{tag}/prev_action", env.action_manager.prev_action)
        # -- physics: feed moves to the next snapshot
        feed.advance()
        for n in in_names:
            put(f"{tag}/in/{n}", feed[n])
        if scanner is not None:  # scene.update(physics_dt) inside the decimation loop (manager_based_rl_env.py:185-196)
            for _ in range(env_cfg.decimation):
                scanner.update(env_cfg.sim.dt)
        else:
            sc = update_scanner()  # RayCaster is lazy; pose does not change on reset in the feed
            if sc is not None:
                put(f"{tag}/ray_starts_w", sc[0])
                put(f"{tag}/ray_hits_w", sc[1])
        # -- post-physics (manager_based_rl_env.py:200-239)
        env.episode_length_buf += 1
        reset_buf = env.termination_manager.compute()
        put(f"{tag}/reset_buf", reset_buf)
        put(f"{tag}/terminated", env.termination_manager.terminated)
        put(f"{tag}/time_outs", env.termination_manager.time_outs)
        for name in env.termination_manager.active_terms:
            put(f"{tag}/term_dones/{name}", env.termination_manager.get_term(name))
        reward = env.reward_manager.compute(dt=env.step_dt)
        put(f"{tag}/reward", reward)
        put(f"{tag}/step_reward", env.reward_manager._step_reward)
        for name in env.reward_manager.active_terms:
            put(f"{tag}/episode_sums_pre_reset/{name}", env.reward_manager._episode_sums[name])
        reset_env_ids = reset_buf.nonzero(as_tuple=False).squeeze(-1)
        put(f"{tag}/reset_env_ids", reset_env_ids)
        log = {}
        if len(reset_env_ids) > 0:
            log.update(env.observation_manager.reset(reset_env_ids))
            log.update(env.action_manager.reset(reset_env_ids))
            log.update(env.reward_manager.reset(reset_env_ids))
            log.update(env.termination_manager.reset(reset_env_ids))
            env.episode_length_buf[reset_env_ids] = 0
            if scanner is not None:  # _reset_idx -> scene.reset(env_ids)
                scanner.reset(reset_env_ids)
        if scanner is not None:
            put(f"{tag}/scan_drift", scanner.drift)
            sc = update_scanner()
            put(f"{tag}/ray_hits_fresh", sc[1])
        rec[f"{tag}/log_json"] = np.array(json.dumps({k: float(v) for k, v in log.items()}))
        for name in env.reward_manager.active_terms:
            put(f"{tag}/episode_sums/{name}", env.reward_manager._episode_sums[name])
        put(f"{tag}/episode_length_buf", env.episode_length_buf)
        put(f"{tag}/action_after_reset", env.action_manager.action)
        put(f"{tag}/prev_action_after_reset", env.action_manager.prev_action)
        compute_obs(tag)

    if kitchen:
        meta["weight_changes"] = [list(w) for w in kitchen.get("weight_changes", ())]
        meta["real_scanner"] = bool(kitchen.get("real_scanner"))
    rec["meta_json"] = np.array(json.dumps(meta))
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, task + ".npz"), **rec)
    print(f"[golden] {task}: N={N} D={D} A={A} steps={steps} reward_terms={meta['reward_terms']}")


# --------------------------------------------------------------------------- math / mesh / pattern fixtures
def math_fixture():
    g = torch.Generator().manual_seed(7)
    q = torch.randn(1024, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    edge = torch.tensor([[1.0, 0, 0, 0], [0, 0, 0, 1.0], [0.70710678, 0, 0, 0.70710678], [0.5, 0.5, 0.5, 0.5],
                         [0, 1.0, 0, 0], [0.70710678, 0.70710678, 0, 0]])
    q = torch.cat([q, edge], 0)
    v = torch.randn(q.shape[0], 3, generator=g) * 3.0
    ang = torch.cat([torch.randn(1000, generator=g) * 10.0,
                     torch.tensor([math.pi, -math.pi, 3 * math.pi, -3 * math.pi, 0.0, 2 * math.pi, 1e-8])])
    out = dict(q=q, v=v, ang=ang,
               quat_rotate_inverse=ref_math.quat_rotate_inverse(q, v), quat_rotate=ref_math.quat_rotate(q, v),
               quat_apply=ref_math.quat_apply(q, v), yaw_quat=ref_math.yaw_quat(q),
               quat_apply_yaw=ref_math.quat_apply_yaw(q, v), wrap_to_pi=ref_math.wrap_to_pi(ang),
               convert_quat_to_wxyz=ref_math.convert_quat(q, to="wxyz"),
               convert_quat_to_xyzw=ref_math.convert_quat(q, to="xyzw"), normalize=ref_math.normalize(v))
    lo = torch.randn(q.shape[0], 3, generator=g) - 2
    hi = lo + torch.rand(q.shape[0], 3, generator=g) * 4 + 0.1
    out.update(lower=lo, upper=hi, scale_transform=ref_math.scale_transform(v, lo, hi))
    np.savez_compressed(os.path.join(GOLDEN, "math.npz"), **{k: t.numpy() for k, t in out.items()})
    print("[golden] math")


def mesh_fixture():
    rng = np.random.default_rng(3)
    hf = np.rint(rng.uniform(0, 40, size=(21, 17)))
    hf[5:9, 4:8] += 60  # a box with vertical walls so that slope snapping triggers
    v0, t0 = convert_height_field_to_mesh(hf, 0.1, 0.005, None)
    v1, t1 = convert_height_field_to_mesh(hf, 0.1, 0.005, 0.75)
    cfg = types.SimpleNamespace(resolution=0.1, size=[1.6, 1.0], direction=(0.0, 0.0, -1.0), ordering="xy")
    s, d = ref_patterns.grid_pattern(cfg, "cpu")
    cfg2 = types.SimpleNamespace(resolution=0.25, size=[1.0, 0.5], direction=(0.0, 0.0, -1.0), ordering="yx")
    s2, d2 = ref_patterns.grid_pattern(cfg2, "cpu")
    np.savez_compressed(os.path.join(GOLDEN, "hf_mesh.npz"), hf=hf, v_none=v0, t_none=t0.astype(np.int64), v_thr=v1,
                        t_thr=t1.astype(np.int64), grid_xy_starts=s.numpy(), grid_xy_dirs=d.numpy(),
                        grid_yx_starts=s2.numpy(), grid_yx_dirs=d2.numpy())
    print("[golden] hf_mesh + grid_pattern", s.shape, s2.shape)


KITCHEN = "Isaac-Velocity-Rough-Anymal-C-v0-kitchen"


def kitchen_cfg():
    """The rough Anymal-C task with every remaining op of isaaclab.envs.mdp on top (SURVEY.md 8a "also present" rows): the cheap
    observation terms, the optional reward and termination terms, a second ("critic") observation group that scans the same height
    scanner, and a scanner with a drift range (its update period stays the task's 0.02 s)."""
    import isaaclab.envs.mdp as mdp
    import isaaclab_tasks.manager_based.locomotion.velocity.mdp as vel_mdp
    from isaaclab.managers import ObservationGroupCfg, ObservationTermCfg as Obs, RewardTermCfg as Rew, SceneEntityCfg
    from isaaclab.managers import TerminationTermCfg as Done
    from isaaclab.utils.noise import AdditiveUniformNoiseCfg as Unoise
    from isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.rough_env_cfg import AnymalCRoughEnvCfg

    cfg = AnymalCRoughEnvCfg()
    robot = lambda **kw: SceneEntityCfg("robot", **kw)  # noqa: E731
    pol = cfg.observations.policy
    pol.base_pos_z = Obs(func=mdp.base_pos_z, noise=Unoise(n_min=-0.01, n_max=0.01))
    pol.root_pos = Obs(func=mdp.root_pos_w)
    pol.root_quat_unique = Obs(func=mdp.root_quat_w, params={"make_quat_unique": True})
    pol.root_lin_vel_w = Obs(func=mdp.root_lin_vel_w, noise=Unoise(n_min=-0.05, n_max=0.05), clip=(-1.0, 1.0))
    pol.root_ang_vel_w = Obs(func=mdp.root_ang_vel_w, scale=0.25)
    pol.haa_pos = Obs(func=mdp.joint_pos, params={"asset_cfg": robot(joint_names=[".*HAA"])})
    pol.joint_vel_abs = Obs(func=mdp.joint_vel)
    pol.kfe_pos_norm = Obs(func=mdp.joint_pos_limit_normalized, params={"asset_cfg": robot(joint_names=[".*KFE"])},
                           noise=Unoise(n_min=-0.02, n_max=0.02))

    critic = ObservationGroupCfg()
    critic.enable_corruption = False
    critic.concatenate_terms = True
    critic.base_lin_vel = Obs(func=mdp.base_lin_vel, noise=Unoise(n_min=-0.1, n_max=0.1))  # noise dropped: corruption is off
    critic.base_ang_vel = Obs(func=mdp.base_ang_vel)
    critic.projected_gravity = Obs(func=mdp.projected_gravity)
    critic.velocity_commands = Obs(func=mdp.generated_commands, params={"command_name": "base_velocity"})
    critic.joint_pos = Obs(func=mdp.joint_pos_rel)
    critic.joint_vel = Obs(func=mdp.joint_vel_rel, scale=0.05)
    critic.actions = Obs(func=mdp.last_action)
    critic.height_scan = Obs(func=mdp.height_scan, params={"sensor_cfg": SceneEntityCfg("height_scanner"), "offset": 0.3}, clip=(-2.0, 2.0))
    critic.root_quat = Obs(func=mdp.root_quat_w)
    cfg.observations.critic = critic

    rew = cfg.rewards
    rew.dof_pos_limits.weight = -1.0  # a zero-weight term of the task switched on; flat_orientation_l2 stays at 0 (skipped)
    rew.alive = Rew(func=mdp.is_alive, weight=0.3)
    rew.terminating = Rew(func=mdp.is_terminated, weight=-2.0)
    rew.termination_kinds = Rew(func=mdp.is_terminated_term, weight=-3.0, params={"term_keys": ["base_contact", "bad_orient.*"]})
    rew.base_height = Rew(func=mdp.base_height_l2, weight=-1.0, params={"target_height": 0.6})
    rew.body_acc = Rew(func=mdp.body_lin_acc_l2, weight=-1.0e-3, params={"asset_cfg": robot(body_names=".*SHANK")})
    rew.hfe_vel_l2 = Rew(func=mdp.joint_vel_l2, weight=-1.0e-3, params={"asset_cfg": robot(joint_names=".*HFE")})
    rew.joint_vel_l1 = Rew(func=mdp.joint_vel_l1, weight=-1.0e-3, params={"asset_cfg": robot()})
    rew.haa_deviation = Rew(func=mdp.joint_deviation_l1, weight=-0.05, params={"asset_cfg": robot(joint_names=".*HAA")})
    rew.vel_limits = Rew(func=mdp.joint_vel_limits, weight=-0.5, params={"soft_ratio": 0.9})
    rew.torque_limits = Rew(func=mdp.applied_torque_limits, weight=-0.01)
    rew.action_l2 = Rew(func=mdp.action_l2, weight=-0.005)
    rew.thigh_forces = Rew(func=mdp.contact_forces, weight=-0.02,
                           params={"sensor_cfg": SceneEntityCfg("contact_forces", body_names=".*THIGH"), "threshold": 5.0})

    term = cfg.terminations
    term.bad_orientation = Done(func=mdp.bad_orientation, params={"limit_angle": 0.4})
    term.too_low = Done(func=mdp.root_height_below_minimum, params={"minimum_height": 0.54})
    term.haa_vel_limit = Done(func=mdp.joint_vel_out_of_limit, params={"asset_cfg": robot(joint_names="LF_HAA")})
    term.kfe_vel_manual = Done(func=mdp.joint_vel_out_of_manual_limit,
                               params={"max_velocity": 2.5, "asset_cfg": robot(joint_names=["RH_KFE", "LH_KFE"])})
    term.effort_limit = Done(func=mdp.joint_effort_out_of_limit, params={"asset_cfg": robot(joint_names="LF_HFE")})
    term.out_of_bounds = Done(func=vel_mdp.terrain_out_of_bounds, params={"distance_buffer": 57.0}, time_out=True)
    term.cmd_resample = Done(func=mdp.command_resample, params={"command_name": "base_velocity", "num_resamples": 1}, time_out=True)

    cfg.scene.height_scanner.drift_range = (-0.03, 0.03)
    return cfg


def kitchen_feed_tweak(feed: StateFeed):
    """Make the synthetic feed exercise the added ops: quaternions with a negative real part (quat_unique), one joint whose applied
    torque mostly differs from the computed one (joint_effort_out_of_limit is `any(isclose(computed, applied))`), incl. pairs just
    inside / outside torch.isclose's tolerance."""
    q = feed._stack["root_quat_w"]
    q[:, ::5] = -q[:, ::5]
    j = feed.robot.joint_names.index("LF_HFE")
    a, c = feed._stack["applied_torque"], feed._stack["computed_torque"]
    N = feed.num_envs
    a[:, :, j] = c[:, :, j] + 1.0
    a[:, 3::17, j] = c[:, 3::17, j]                       # exactly equal
    a[:, 5::19, j] = c[:, 5::19, j] * (1.0 + 0.9e-5)      # inside rtol 1e-5 (for |c| not tiny)
    a[:, 7::23, j] = c[:, 7::23, j] * (1.0 + 1.3e-5)      # outside
    assert N >= 24


def main():
    from isaaclab_tasks.manager_based.classic.cartpole.agents.rsl_rl_ppo_cfg import CartpolePPORunnerCfg
    from isaaclab_tasks.manager_based.classic.cartpole.cartpole_env_cfg import CartpoleEnvCfg
    from isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.agents.rsl_rl_ppo_cfg import (
        AnymalCFlatPPORunnerCfg,
        AnymalCRoughPPORunnerCfg,
    )
    from isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.flat_env_cfg import AnymalCFlatEnvCfg
    from isaaclab_tasks.manager_based.locomotion.velocity.config.anymal_c.rough_env_cfg import AnymalCRoughEnvCfg
    from isaaclab_tasks.manager_based.locomotion.velocity.config.g1.agents.rsl_rl_ppo_cfg import G1RoughPPORunnerCfg
    from isaaclab_tasks.manager_based.locomotion.velocity.config.g1.rough_env_cfg import G1RoughEnvCfg

    only = set(sys.argv[1:])  # `python oracle/gen_golden.py <task> ...` regenerates just those fixtures

    def want(name):
        return not only or name in only

    if not only:
        math_fixture()
        mesh_fixture()
    verts, tris, ext = make_rough_terrain(**ROUGH_TERRAIN_ARGS)
    mesh = (verts, tris)
    if want("Isaac-Cartpole-v0"):
        run_task("Isaac-Cartpole-v0", CartpoleEnvCfg(), CartpolePPORunnerCfg(), CARTPOLE, N=64, steps=3, seed=101)
    if want("Isaac-Velocity-Flat-Anymal-C-v0"):
        run_task("Isaac-Velocity-Flat-Anymal-C-v0", AnymalCFlatEnvCfg(), AnymalCFlatPPORunnerCfg(), ANYMAL_C, N=64,
                 steps=3, seed=102)
    if want("Isaac-Velocity-Flat-Anymal-C-v0-hist3"):
        # observation history (ObservationGroupCfg.history_length, CircularBuffer): the flat task, 3-deep flattened history
        hist_cfg = AnymalCFlatEnvCfg()
        hist_cfg.observations.policy.history_length = 3
        hist_cfg.observations.policy.flatten_history_dim = True
        run_task("Isaac-Velocity-Flat-Anymal-C-v0-hist3", hist_cfg, AnymalCFlatPPORunnerCfg(), ANYMAL_C, N=64, steps=5, seed=105)
    if want("Isaac-Velocity-Flat-Anymal-C-v0-shapes"):
        # the group shapes besides the flat concatenation (observation_manager.py:320-335): a group returned as a dict of terms
        # (concatenate_terms=False) holding an un-flattened (N, H, d) history term, a flattened one and a plain one; and a concatenated
        # group whose terms keep their history axis (group-level history_length, flatten_history_dim=False): (N, H, sum d)
        import isaaclab.envs.mdp as mdp
        from isaaclab.managers import ObservationGroupCfg, ObservationTermCfg as Obs
        from isaaclab.utils.noise import AdditiveUniformNoiseCfg as Unoise

        shp_cfg = AnymalCFlatEnvCfg()
        terms = ObservationGroupCfg()
        terms.concatenate_terms = False
        terms.enable_corruption = True
        terms.base_lin_vel = Obs(func=mdp.base_lin_vel, noise=Unoise(n_min=-0.1, n_max=0.1), history_length=2, flatten_history_dim=False)
        terms.joint_pos = Obs(func=mdp.joint_pos_rel, noise=Unoise(n_min=-0.01, n_max=0.01))
        terms.actions = Obs(func=mdp.last_action, history_length=3)
        terms.joint_vel = Obs(func=mdp.joint_vel_rel, scale=0.05, clip=(-1.0, 1.0), history_length=2, flatten_history_dim=False)
        shp_cfg.observations.terms = terms
        stack = ObservationGroupCfg()
        stack.concatenate_terms = True
        stack.enable_corruption = False
        stack.history_length = 2
        stack.flatten_history_dim = False
        stack.base_ang_vel = Obs(func=mdp.base_ang_vel)
        stack.projected_gravity = Obs(func=mdp.projected_gravity)
        stack.velocity_commands = Obs(func=mdp.generated_commands, params={"command_name": "base_velocity"})
        shp_cfg.observations.stack = stack
        run_task("Isaac-Velocity-Flat-Anymal-C-v0-shapes", shp_cfg, AnymalCFlatPPORunnerCfg(), ANYMAL_C, N=64, steps=5, seed=108)
    if want("Isaac-Velocity-Flat-Anymal-C-v0-noise"):
        # the noise functions besides uniform_noise (utils/noise/noise_model.py:17-39,71-94): constant and gaussian, each with add / scale / abs
        from isaaclab.utils.noise import ConstantNoiseCfg, GaussianNoiseCfg, UniformNoiseCfg

        nz_cfg = AnymalCFlatEnvCfg()
        pol = nz_cfg.observations.policy
        pol.base_lin_vel.noise = GaussianNoiseCfg(mean=0.01, std=0.1)
        pol.base_ang_vel.noise = ConstantNoiseCfg(bias=0.05)
        pol.projected_gravity.noise = GaussianNoiseCfg(mean=1.0, std=0.05, operation="scale")
        pol.velocity_commands.noise = ConstantNoiseCfg(bias=0.25, operation="abs")
        pol.joint_pos.noise = UniformNoiseCfg(n_min=0.9, n_max=1.1, operation="scale")
        pol.joint_vel.noise = ConstantNoiseCfg(bias=0.9, operation="scale")
        pol.actions.noise = GaussianNoiseCfg(mean=-0.2, std=0.3, operation="abs")
        run_task("Isaac-Velocity-Flat-Anymal-C-v0-noise", nz_cfg, AnymalCFlatPPORunnerCfg(), ANYMAL_C, N=64, steps=3, seed=109)
    if want("Isaac-Velocity-Flat-Anymal-C-v0-actions"):
        # the other joint action terms (envs/mdp/actions/joint_actions.py:163-193, joint_actions_to_limits.py:25-139) and per-joint
        # scale / clip dicts: three terms side by side in one ActionManager
        import isaaclab.envs.mdp as mdp

        act_cfg = AnymalCFlatEnvCfg()
        del act_cfg.actions.joint_pos
        act_cfg.actions.haa_rel = mdp.RelativeJointPositionActionCfg(asset_name="robot", joint_names=[".*HAA"], scale=0.3, offset=0.7)
        act_cfg.actions.hfe_lim = mdp.JointPositionToLimitsActionCfg(asset_name="robot", joint_names=[".*HFE"], scale={".*F_HFE": 0.8, ".*H_HFE": 0.4},
                                                                      clip={"LF_HFE": (-0.5, 0.25)})
        act_cfg.actions.kfe_vel = mdp.JointVelocityActionCfg(asset_name="robot", joint_names=["L.*KFE"], scale=2.0, use_default_offset=False,
                                                             offset={"LF.*": 0.1, "LH.*": -0.1}, clip={"LH_KFE": (-1.5, 1.5)})
        # (over ALL joints: the reference's EMA reset indexes with env_ids[:, None] and raises for a joint subset, joint_actions_to_limits.py:206)
        act_cfg.actions.all_ema = mdp.EMAJointPositionToLimitsActionCfg(asset_name="robot", joint_names=[".*"], scale=0.9,
                                                                        alpha={".*HAA": 0.3, ".*HFE": 0.75, ".*KFE": 1.0})
        run_task("Isaac-Velocity-Flat-Anymal-C-v0-actions", act_cfg, AnymalCFlatPPORunnerCfg(), ANYMAL_C, N=64, steps=3, seed=110)
    if want("Isaac-Velocity-Flat-Anymal-C-v0-mod"):
        # observation modifiers (ObservationTermCfg.modifiers; utils/modifiers/modifier.py): stateless chain, IIR/FIR filter,
        # integrator -- on the flat task, 6 steps so that filter/integrator state and its reset are exercised
        from isaaclab.utils import modifiers as ref_mod

        mod_cfg = AnymalCFlatEnvCfg()
        pol = mod_cfg.observations.policy
        pol.base_lin_vel.modifiers = [ref_mod.ModifierCfg(func=ref_mod.scale, params={"multiplier": 2.0}),
                                      ref_mod.ModifierCfg(func=ref_mod.bias, params={"value": 0.25}),
                                      ref_mod.ModifierCfg(func=ref_mod.clip, params={"bounds": (-0.8, None)})]
        pol.base_ang_vel.modifiers = [ref_mod.DigitalFilterCfg(A=[0.0], B=[0.0, 1.0])]  # unit delay
        pol.joint_pos.modifiers = [ref_mod.IntegratorCfg(dt=0.02), ref_mod.ModifierCfg(func=ref_mod.clip, params={"bounds": (-0.01, 0.015)})]
        pol.joint_vel.modifiers = [ref_mod.DigitalFilterCfg(A=[-0.5, 0.1], B=[0.3, 0.2, 0.1]),
                                   ref_mod.ModifierCfg(func=ref_mod.scale, params={"multiplier": 0.5})]
        pol.actions.modifiers = [ref_mod.DigitalFilterCfg(A=[0.6], B=[0.4])]  # first-order low-pass
        run_task("Isaac-Velocity-Flat-Anymal-C-v0-mod", mod_cfg, AnymalCFlatPPORunnerCfg(), ANYMAL_C, N=64, steps=6, seed=106)
    if want("Isaac-Velocity-Rough-Anymal-C-v0"):
        run_task("Isaac-Velocity-Rough-Anymal-C-v0", AnymalCRoughEnvCfg(), AnymalCRoughPPORunnerCfg(), ANYMAL_C, N=64,
                 steps=3, seed=103, mesh=mesh, extent=(ext[0] - 1.0, ext[1] - 1.0))
    if want(KITCHEN):
        ts0 = torch.tensor([0.0, 16.5, 30.0, 5.0]).repeat(16)  # fp32 timestamps of fresh, 16 s, 30 s and 5 s old sensors
        run_task(KITCHEN, kitchen_cfg(), AnymalCRoughPPORunnerCfg(), ANYMAL_C, N=64, steps=5, seed=107, mesh=mesh,
                 extent=(ext[0] - 1.0, ext[1] - 1.0),
                 kitchen=dict(real_scanner=True, scan_ts0=ts0, feed_tweak=kitchen_feed_tweak,
                              weight_changes=[(2, "action_l2", -0.02), (3, "flat_orientation_l2", -1.5), (4, "alive", 0.0)]))
    if want("Isaac-Velocity-Rough-G1-v0"):
        run_task("Isaac-Velocity-Rough-G1-v0", G1RoughEnvCfg(), G1RoughPPORunnerCfg(), G1, N=64, steps=3, seed=104,
                 mesh=mesh, extent=(ext[0] - 1.0, ext[1] - 1.0))


if __name__ == "__main__":
    main()
