"""Term compiler: a reference env cfg (live ``@configclass`` object or its ``to_dict()``/JSON form) -> libimx plan blob.

Replaces the init-time work of the reference managers -- ``ManagerBase._resolve_common_term_cfg`` /
``_process_term_cfg_at_play`` (isaaclab/managers/manager_base.py:278-395), ``SceneEntityCfg.resolve``
(managers/scene_entity_cfg.py:112-250), ``ObservationManager._prepare_terms`` (observation_manager.py:337-470),
``RewardManager._prepare_terms`` (reward_manager.py:211-250), ``TerminationManager._prepare_terms``
(termination_manager.py:198-230), ``ActionManager._prepare_terms`` (action_manager.py:365-393) -- and keys the fused
ops on the *qualified function name* of each term (``"module:function"``, the form ``configclass.to_dict`` emits,
isaaclab/utils/dict.py:23-72).  Terms whose function is unknown are routed to ``IMX_*_EXTERNAL`` and evaluated by
calling the Python term (correct, slow) -- see ``env.py``.
"""

from __future__ import annotations

import dataclasses
import math
import struct
from typing import Any, Callable

import numpy as np

from .robots import RobotSpec, resolve_matching_names, resolve_matching_names_values

# ---- constants mirrored from include/imx.h (tests/test_boundary.py checks they agree) --------------------------
MAGIC = 0x31584D49
PLAN_VERSION = 3
HEADER_WORDS = 48
MAX_OBS_GROUPS = 4
REC_WORDS = 20
H = dict(MAGIC=0, VERSION=1, J=2, B=3, H=4, A=5, D=6, R=7, NTERM=8, NREW=9, NOBS=10, NACT=11, MAX_EP_LEN=12,
         STEP_DT=13, TERM_OFF=14, REW_OFF=15, OBS_OFF=16, ACT_OFF=17, TOTAL_WORDS=18, NB=19, GRAV_X=20, GRAV_Y=21,
         GRAV_Z=22, NREW_ALL=23, RAY_OFF=24, RAYDIR_X=25, RAYDIR_Y=26, RAYDIR_Z=27, RAY_MAXDIST=28, MAX_EP_LEN_S=29,
         NEXT_REW=30, NEXT_TERM=31, NEXT_OBS=32, RAY_YAW_ONLY=33, CMD_DIM=34, MOD_STATE=35, NGROUPS=36, GROUP_OFF=37,
         SCAN_PERIOD=38, SCAN_DT=39, SCAN_SUBSTEPS=40, SCAN_DRIFT_LO=41, SCAN_DRIFT_HI=42, SCAN_STATEFUL=43)
R = dict(OP=0, IDS_OFF=1, NIDS=2, IDS2_OFF=3, NIDS2=4, WEIGHT=5, P0=6, P1=7, P2=8, P3=9, OUT=10, DIM=11, FLAGS=12,
         NOISE_LO=13, NOISE_HI=14, CLIP_LO=15, CLIP_HI=16, SCALE=17, AUX0=18, AUX1=19)
F_NOISE_ADD, F_NOISE_SCALE, F_NOISE_ABS, F_CLIP, F_SCALE, F_QUAT_UNIQUE, F_MODIFIERS, F_SCAN_TWIN = 1, 2, 4, 8, 16, 32, 64, 128
F_NOISE_GAUSS = 1024
F_ACT_TO_LIMITS = 8
F_ACT_EMA = 16
M_OPS = dict(SCALE=1, BIAS=2, CLIP=3, INTEGRATOR=4, DIGITAL_FILTER=5)
F_ACT_DEFAULT_POS_OFFSET, F_ACT_DEFAULT_VEL_OFFSET, F_ACT_CLIP = 1, 2, 4

T_OPS = dict(TIME_OUT=1, ILLEGAL_CONTACT=2, JOINT_POS_MANUAL_LIMIT=3, BAD_ORIENTATION=4, ROOT_HEIGHT_BELOW_MIN=5,
             JOINT_VEL_LIMIT=6, JOINT_VEL_MANUAL_LIMIT=7, JOINT_EFFORT_LIMIT=8, TERRAIN_OUT_OF_BOUNDS=9, EXTERNAL=10,
             COMMAND_RESAMPLE=11)
W_OPS = dict(IS_ALIVE=1, IS_TERMINATED=2, IS_TERMINATED_TERM=3, LIN_VEL_Z_L2=4, ANG_VEL_XY_L2=5, FLAT_ORIENTATION_L2=6,
             BASE_HEIGHT_L2=7, JOINT_TORQUES_L2=8, JOINT_VEL_L1=9, JOINT_VEL_L2=10, JOINT_ACC_L2=11,
             JOINT_DEVIATION_L1=12, JOINT_POS_LIMITS=13, JOINT_VEL_LIMITS=14, APPLIED_TORQUE_LIMITS=15,
             ACTION_RATE_L2=16, ACTION_L2=17, UNDESIRED_CONTACTS=18, CONTACT_FORCES=19, TRACK_LIN_VEL_XY_EXP=20,
             TRACK_ANG_VEL_Z_EXP=21, FEET_AIR_TIME=22, FEET_AIR_TIME_POSITIVE_BIPED=23, FEET_SLIDE=24,
             TRACK_LIN_VEL_XY_YAW_FRAME_EXP=25, TRACK_ANG_VEL_Z_WORLD_EXP=26, JOINT_POS_TARGET_L2=27, EXTERNAL=28,
             BODY_LIN_ACC_L2=29)
O_OPS = dict(BASE_POS_Z=1, BASE_LIN_VEL=2, BASE_ANG_VEL=3, PROJECTED_GRAVITY=4, ROOT_POS_W=5, ROOT_QUAT_W=6,
             ROOT_LIN_VEL_W=7, ROOT_ANG_VEL_W=8, JOINT_POS=9, JOINT_POS_REL=10, JOINT_POS_LIMIT_NORMALIZED=11,
             JOINT_VEL=12, JOINT_VEL_REL=13, HEIGHT_SCAN=14, LAST_ACTION=15, GENERATED_COMMANDS=16, EXTERNAL=17)
A_JOINT_AFFINE = 1

_MDP = "isaaclab.envs.mdp"
_VEL = "isaaclab_tasks.manager_based.locomotion.velocity.mdp"
_CART = "isaaclab_tasks.manager_based.classic.cartpole.mdp"


def f32(x: float) -> float:
    """Python scalar -> the float32 value torch uses when it meets a float32 tensor."""
    return float(np.float32(x))


def _f2w(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def func_name(func: Any) -> str:
    """``module:function`` of a term function given as string or callable (isaaclab/utils/string.py:108-135)."""
    if isinstance(func, str):
        return func
    mod = getattr(func, "__module__", None)
    name = getattr(func, "__qualname__", getattr(func, "__name__", None))
    return f"{mod}:{name}"


def _short(name: str) -> tuple[str, str]:
    mod, _, fn = name.partition(":")
    return mod, fn


def _to_dict(cfg: Any) -> Any:
    if isinstance(cfg, dict):
        return cfg
    if hasattr(cfg, "to_dict"):
        return cfg.to_dict()
    raise TypeError(f"expected a configclass instance or its dict form, got {type(cfg)}")


def _is_slice_all(x) -> bool:
    return x is None or x == slice(None) or (isinstance(x, str) and x.replace(" ", "") == "slice(None,None,None)")


@dataclasses.dataclass
class Term:
    name: str
    func: str
    op: int
    params: dict
    external: Callable | None = None  # python fallback
    dim: int = 1
    weight: float = 0.0
    time_out: bool = False
    py_modifiers: list = dataclasses.field(default_factory=list)  # (func, params) of a foreign modifier chain applied in Python


@dataclasses.dataclass
class Plan:
    blob: np.ndarray  # int32 words
    robot: RobotSpec
    num_joints: int
    num_bodies: int
    history: int
    action_dim: int
    obs_dim: int
    num_rays: int
    cmd_dim: int
    step_dt: float
    max_episode_length: int
    max_episode_length_s: float
    is_finite_horizon: bool
    reward_terms: list[Term]  # ALL reward terms incl. zero weight (active_terms order)
    termination_terms: list[Term]
    obs_terms: list[Term]
    obs_term_dims: list[tuple[int, ...]]
    action_terms: list[Term]
    enable_corruption: bool
    ray_starts_local: np.ndarray | None
    ray_direction: tuple[float, float, float]
    ray_max_distance: float
    scanner_cfg: dict | None
    n_ext_rew: int = 0
    n_ext_term: int = 0
    n_ext_obs: int = 0
    mod_state_dim: int = 0  # floats of observation-modifier state per env (DigitalFilter / Integrator)
    gravity_dir: tuple[float, float, float] = (0.0, 0.0, -1.0)
    obs_groups: list = dataclasses.field(default_factory=list)  # every observation group (ObsGroup), cfg order; [0] = obs_terms/obs_dim
    obs_dim_total: int = 0  # sum of the group widths (= width of the parity-mode noise feed)
    scan_stateful: bool = False  # the height scanner keeps per-env timestamps / drift (update_period > 0 or a drift range)
    scan_drift_range: tuple[float, float] = (0.0, 0.0)


@dataclasses.dataclass
class ObsGroup:
    name: str
    enable_corruption: bool = False
    first_record: int = 0
    num_records: int = 0
    dim: int = 0
    terms: list = dataclasses.field(default_factory=list)
    term_dims: list = dataclasses.field(default_factory=list)    # the reference's group_obs_term_dim: (d,), (H*d,) or (H, d) per term
    term_widths: list = dataclasses.field(default_factory=list)  # columns of each term in the fused row (history windows flattened, oldest first)
    concatenate: bool = True                                     # ObservationGroupCfg.concatenate_terms

    @property
    def is_flat(self) -> bool:
        """The group IS its fused row: concatenated, every term one-dimensional."""
        return self.concatenate and all(len(d) == 1 for d in self.term_dims)


class _Blob:
    def __init__(self):
        self.w: list[int] = [0] * HEADER_WORDS

    def ints(self, xs) -> int:
        off = len(self.w)
        self.w.extend(int(x) for x in xs)
        return off

    def floats(self, xs) -> int:
        off = len(self.w)
        self.w.extend(_f2w(x) for x in xs)
        return off

    def table(self, recs: list[list[int]]) -> int:
        off = len(self.w)
        for r in recs:
            assert len(r) == REC_WORDS
            self.w.extend(r)
        return off


def compile_modifiers(mods) -> tuple[list[int], int]:
    """``ObservationTermCfg.modifiers`` (isaaclab/utils/modifiers/modifier_cfg.py; applied at observation_manager.py:310-312)
    -> (program words, state slots per element) in the format of include/imx.h ``IMX_F_MODIFIERS``.  Raises
    NotImplementedError for a modifier that is not one of modifier.py's five."""
    prog: list[int] = []
    slots = 0
    for m in mods:
        m = m if isinstance(m, dict) else m.to_dict()
        mod, fn = _short(func_name(m["func"]))
        if not mod.startswith("isaaclab.utils.modifiers"):
            raise NotImplementedError(f"modifier {mod}:{fn}")
        p = m.get("params") or {}
        if fn == "scale":
            prog += [M_OPS["SCALE"], _f2w(f32(p["multiplier"])), 0, 0]
        elif fn == "bias":
            prog += [M_OPS["BIAS"], _f2w(f32(p["value"])), 0, 0]
        elif fn == "clip":
            lo, hi = p["bounds"]
            prog += [M_OPS["CLIP"], _f2w(-math.inf if lo is None else f32(lo)), _f2w(math.inf if hi is None else f32(hi)), 0]
        elif fn == "Integrator":
            prog += [M_OPS["INTEGRATOR"], _f2w(f32(m["dt"])), 0, slots]
            slots += 2
        elif fn == "DigitalFilter":
            A_, B_ = m.get("A"), m.get("B")
            if A_ is None or B_ is None:  # modifier.py:131-132
                raise ValueError("Digital filter coefficients A and B must not be None. Please provide valid coefficients.")
            prog += [M_OPS["DIGITAL_FILTER"], len(A_), len(B_), slots] + [_f2w(f32(x)) for x in list(A_) + list(B_)]
            slots += len(A_) + len(B_)
        else:
            raise NotImplementedError(f"modifier {mod}:{fn}")
    return prog, slots


def _rec(**kw) -> list[int]:
    r = [0] * REC_WORDS
    for k, v in kw.items():
        key = k.upper()
        r[R[key]] = _f2w(v) if isinstance(v, float) else int(v)
    return r


def grid_pattern(resolution: float, size, direction=(0.0, 0.0, -1.0), ordering: str = "xy"):
    """``grid_pattern`` (isaaclab/sensors/ray_caster/patterns/patterns.py:16-58).  ``torch.arange`` on float32
    evaluates ``start + i*step`` in double and rounds to float32; ``meshgrid`` 'xy' puts x fastest."""
    if ordering not in ("xy", "yx"):
        raise ValueError(f"Ordering must be 'xy' or 'yx'. Received: '{ordering}'.")
    if resolution <= 0:
        raise ValueError(f"Resolution must be greater than 0. Received: '{resolution}'.")

    def arange(start, end, step):
        n = int(math.ceil((end - start) / step))
        return (start + step * np.arange(n, dtype=np.float64)).astype(np.float32)

    x = arange(-size[0] / 2, size[0] / 2 + 1.0e-9, resolution)
    y = arange(-size[1] / 2, size[1] / 2 + 1.0e-9, resolution)
    if ordering == "xy":  # torch.meshgrid(indexing="xy"): output shape (len(y), len(x))
        gx, gy = np.meshgrid(x, y, indexing="xy")
    else:  # "ij"
        gx, gy = np.meshgrid(x, y, indexing="ij")
    starts = np.zeros((gx.size, 3), np.float32)
    starts[:, 0] = gx.reshape(-1)
    starts[:, 1] = gy.reshape(-1)
    dirs = np.tile(np.asarray(direction, np.float32), (gx.size, 1))
    return starts, dirs


def _quat_apply_np(q, v):
    w, xyz = np.float32(q[0]), np.asarray(q[1:], np.float32)
    t = np.cross(xyz, v).astype(np.float32) * np.float32(2)
    return (v + w * t + np.cross(xyz, t)).astype(np.float32)


class PlanCompiler:
    """Compile one env cfg.  ``entities`` maps scene entity names to name tables."""

    def __init__(self, env_cfg: Any, robot: RobotSpec, external_env_getter: Callable | None = None):
        self.cfg = _to_dict(env_cfg)
        self.live_cfg = None if isinstance(env_cfg, dict) else env_cfg
        self.robot = robot
        self.joint_names = list(robot.joint_names)
        self.body_names = list(robot.body_names)

    # -- SceneEntityCfg.resolve -------------------------------------------------------------------------------
    def _entity_names(self, entity: str, kind: str) -> list[str]:
        if entity in ("robot", "contact_forces"):
            return self.joint_names if kind == "joint" else self.body_names
        raise ValueError(f"The scene entity '{entity}' does not exist. Available entities: "
                         f"['robot', 'contact_forces', 'height_scanner'].")

    def resolve_ids(self, ent: Any, kind: str, default_entity: str = "robot") -> list[int]:
        """joint_ids / body_ids of a SceneEntityCfg (dict form, live object, or None = function default)."""
        if ent is None:
            return list(range(len(self._entity_names(default_entity, kind))))
        get = (lambda k: ent.get(k)) if isinstance(ent, dict) else (lambda k: getattr(ent, k, None))
        name = get("name")
        names = self._entity_names(name, kind)
        keys, ids = get(f"{kind}_names"), get(f"{kind}_ids")
        preserve = bool(get("preserve_order"))
        if keys is not None and not _is_slice_all(ids):
            if isinstance(keys, str):
                keys = [keys]
            if isinstance(ids, int):
                ids = [ids]
            r_ids, _ = resolve_matching_names(keys, names, preserve)
            if list(r_ids) != list(ids) or [names[i] for i in ids] != list(keys):
                raise ValueError(f"Both '{kind}_names' and '{kind}_ids' are specified, and are not consistent.")
            return list(ids)
        if keys is not None:
            if isinstance(keys, str):
                keys = [keys]
            r_ids, _ = resolve_matching_names(keys, names, preserve)
            return list(r_ids)
        if not _is_slice_all(ids):
            return [ids] if isinstance(ids, int) else [int(i) for i in ids]
        return list(range(len(names)))

    # -- compile ------------------------------------------------------------------------------------------------
    def compile(self) -> Plan:
        cfg, robot = self.cfg, self.robot
        J, B = robot.num_joints, robot.num_bodies
        scene = cfg.get("scene", {})
        contact = scene.get("contact_forces")
        Hh = int(contact.get("history_length", 0)) if contact else 0
        Hh = max(Hh, 1) if contact else 1
        step_dt = cfg["sim"]["dt"] * cfg["decimation"]
        max_len_s = float(cfg["episode_length_s"])
        max_len = math.ceil(max_len_s / step_dt)  # manager_based_rl_env.py:100-103
        gravity = cfg["sim"].get("gravity", (0.0, 0.0, -9.81))
        g = np.asarray(gravity, np.float32)
        gdir = g / max(float(np.linalg.norm(g)), 1e-9)  # articulation_data.py:54-60
        blob = _Blob()

        # ---- actions (ActionManager._prepare_terms; JointAction.__init__ joint_actions.py:55-112)
        action_terms: list[Term] = []
        act_recs: list[list[int]] = []
        A = 0
        for name, tcfg in (cfg.get("actions") or {}).items():
            if tcfg is None or not isinstance(tcfg, dict) or "class_type" not in tcfg:
                continue
            cls = func_name(tcfg["class_type"])
            _, cname = _short(cls)
            if cname not in ("JointPositionAction", "JointVelocityAction", "JointEffortAction", "RelativeJointPositionAction",
                             "JointPositionToLimitsAction", "EMAJointPositionToLimitsAction"):
                raise NotImplementedError(f"action term '{name}': class {cls} is not on the fused path")
            ids, jn = resolve_matching_names(tcfg["joint_names"], self.joint_names, bool(tcfg.get("preserve_order")))
            dim = len(ids)
            rec = dict(op=A_JOINT_AFFINE, ids_off=blob.ints(ids), nids=dim, out=A, dim=dim)
            flags = 0
            scale, offset = tcfg.get("scale", 1.0), tcfg.get("offset", 0.0)
            if cname.endswith("JointPositionToLimitsAction") or (cname == "RelativeJointPositionAction" and tcfg.get("use_zero_offset", True)):
                offset = 0.0  # joint_actions_to_limits.py:111 (no offset at all), joint_actions.py:180-182
            if cname == "EMAJointPositionToLimitsAction":  # the offset slots carry the moving-average weight (:174-193)
                offset = tcfg.get("alpha", 1.0)
                if isinstance(offset, dict):
                    for jname_, v_ in zip(*resolve_matching_names_values(offset, jn)[1:]):
                        if not 0.0 <= v_ <= 1.0:
                            raise ValueError(f"Moving average weight must be in the range [0, 1]. Got {v_} for joint {jname_}.")
                elif isinstance(offset, float):
                    if not 0.0 <= offset <= 1.0:
                        raise ValueError(f"Moving average weight must be in the range [0, 1]. Got {offset}.")
                else:
                    raise ValueError(f"Unsupported moving average weight type: {type(offset)}. Supported types are float and dict.")
            if not isinstance(scale, (int, float, dict)):
                raise ValueError(f"Unsupported scale type: {type(scale)}. Supported types are float and dict.")
            if not isinstance(offset, (int, float, dict)):
                raise ValueError(f"Unsupported offset type: {type(offset)}. Supported types are float and dict.")
            if isinstance(scale, dict):
                tab = [1.0] * dim
                i_, _, v_ = resolve_matching_names_values(scale, jn)
                for i, v in zip(i_, v_):
                    tab[i] = float(v)
                rec["aux0"] = blob.floats(tab)
            else:
                rec["p0"] = float(scale)
            if isinstance(offset, dict):
                tab = [1.0 if cname == "EMAJointPositionToLimitsAction" else 0.0] * dim
                i_, _, v_ = resolve_matching_names_values(offset, jn)
                for i, v in zip(i_, v_):
                    tab[i] = float(v)
                rec["aux1"] = blob.floats(tab)
            else:
                rec["p1"] = float(offset)
            if cname == "JointPositionAction" and tcfg.get("use_default_offset", True):
                flags |= F_ACT_DEFAULT_POS_OFFSET
            if cname == "JointVelocityAction" and tcfg.get("use_default_offset", True):
                flags |= F_ACT_DEFAULT_VEL_OFFSET
            if cname.endswith("JointPositionToLimitsAction") and tcfg.get("rescale_to_limits", True):
                flags |= F_ACT_TO_LIMITS
            if cname == "EMAJointPositionToLimitsAction":
                flags |= F_ACT_EMA
            if tcfg.get("clip") is not None:
                if not isinstance(tcfg["clip"], dict):
                    raise ValueError(f"Unsupported clip type: {type(tcfg['clip'])}. Supported types are dict.")
                tab = [-math.inf, math.inf] * dim
                i_, _, v_ = resolve_matching_names_values(tcfg["clip"], jn)
                for i, v in zip(i_, v_):
                    tab[2 * i], tab[2 * i + 1] = float(v[0]), float(v[1])
                rec["ids2_off"] = blob.floats(tab)
                rec["nids2"] = 2 * dim
                flags |= F_ACT_CLIP
            rec["flags"] = flags
            act_recs.append(_rec(**rec))
            action_terms.append(Term(name, cls, A_JOINT_AFFINE, dict(tcfg), dim=dim))
            A += dim

        # ---- terminations
        term_terms: list[Term] = []
        term_recs: list[list[int]] = []
        n_ext_term = 0
        for name, tcfg in (cfg.get("terminations") or {}).items():
            if tcfg is None:
                continue
            fn = func_name(tcfg["func"])
            p = dict(tcfg.get("params") or {})
            k = len(term_terms)
            rec = dict(out=k, weight=1 if tcfg.get("time_out") else 0)
            mod, short = _short(fn)
            known = True
            if fn == f"{_MDP}.terminations:time_out":
                rec["op"] = T_OPS["TIME_OUT"]
            elif fn == f"{_MDP}.terminations:illegal_contact":
                ids = self.resolve_ids(p["sensor_cfg"], "body")
                rec.update(op=T_OPS["ILLEGAL_CONTACT"], ids_off=blob.ints(ids), nids=len(ids), p0=f32(p["threshold"]))
            elif fn == f"{_MDP}.terminations:joint_pos_out_of_manual_limit":
                ids = self.resolve_ids(p.get("asset_cfg"), "joint")
                rec.update(op=T_OPS["JOINT_POS_MANUAL_LIMIT"], ids_off=blob.ints(ids), nids=len(ids),
                           p0=f32(p["bounds"][0]), p1=f32(p["bounds"][1]))
            elif fn == f"{_MDP}.terminations:bad_orientation":
                rec.update(op=T_OPS["BAD_ORIENTATION"], p0=f32(p["limit_angle"]))
            elif fn == f"{_MDP}.terminations:root_height_below_minimum":
                rec.update(op=T_OPS["ROOT_HEIGHT_BELOW_MIN"], p0=f32(p["minimum_height"]))
            elif fn == f"{_MDP}.terminations:joint_vel_out_of_limit":
                ids = self.resolve_ids(p.get("asset_cfg"), "joint")
                rec.update(op=T_OPS["JOINT_VEL_LIMIT"], ids_off=blob.ints(ids), nids=len(ids))
            elif fn == f"{_MDP}.terminations:joint_vel_out_of_manual_limit":
                ids = self.resolve_ids(p.get("asset_cfg"), "joint")
                rec.update(op=T_OPS["JOINT_VEL_MANUAL_LIMIT"], ids_off=blob.ints(ids), nids=len(ids),
                           p0=f32(p["max_velocity"]))
            elif fn == f"{_MDP}.terminations:joint_effort_out_of_limit":
                ids = self.resolve_ids(p.get("asset_cfg"), "joint")
                rec.update(op=T_OPS["JOINT_EFFORT_LIMIT"], ids_off=blob.ints(ids), nids=len(ids))
            elif fn == f"{_MDP}.terminations:command_resample":
                # time_left (f32) <= step_dt and command_counter == num_resamples (terminations.py:35-42)
                rec.update(op=T_OPS["COMMAND_RESAMPLE"], p0=f32(step_dt), nids=int(p.get("num_resamples", 1)))
            elif fn == f"{_VEL}.terminations:terrain_out_of_bounds":
                terr = scene.get("terrain") or {}
                if terr.get("terrain_type") == "plane" or not terr.get("terrain_generator"):
                    rec.update(op=T_OPS["TERRAIN_OUT_OF_BOUNDS"], p0=math.inf, p1=math.inf)
                else:
                    tg = terr["terrain_generator"]
                    buf = float(p.get("distance_buffer", 3.0))
                    mw = tg["num_rows"] * tg["size"][0] + 2 * tg["border_width"]
                    mh = tg["num_cols"] * tg["size"][1] + 2 * tg["border_width"]
                    rec.update(op=T_OPS["TERRAIN_OUT_OF_BOUNDS"], p0=f32(0.5 * mw - buf), p1=f32(0.5 * mh - buf))
            else:
                known = False
                rec.update(op=T_OPS["EXTERNAL"], aux0=n_ext_term)
                n_ext_term += 1
            term_recs.append(_rec(**rec))
            term_terms.append(Term(name, fn, rec["op"], p, external=None if known else tcfg["func"],
                                   time_out=bool(tcfg.get("time_out"))))

        # ---- rewards (zero-weight terms keep their slot and record but are skipped at run time: reward_manager.py:145)
        rew_terms: list[Term] = []
        rew_recs: list[list[int]] = []
        n_ext_rew = 0
        for name, tcfg in (cfg.get("rewards") or {}).items():
            if tcfg is None:
                continue
            fn = func_name(tcfg["func"])
            p = dict(tcfg.get("params") or {})
            weight = tcfg["weight"]
            if not isinstance(weight, (float, int)):
                raise TypeError(f"Weight for the term '{name}' is not of type float or int. Received: '{type(weight)}'.")
            idx = len(rew_terms)
            rec: dict[str, Any] = dict(out=idx, weight=f32(weight))

            def joints(key="asset_cfg"):
                ids = self.resolve_ids(p.get(key), "joint")
                rec.update(ids_off=blob.ints(ids), nids=len(ids))

            def bodies(key="sensor_cfg"):
                ids = self.resolve_ids(p.get(key), "body", "contact_forces")
                rec.update(ids_off=blob.ints(ids), nids=len(ids))

            known = True
            table = {
                f"{_MDP}.rewards:is_alive": ("IS_ALIVE", None), f"{_MDP}.rewards:is_terminated": ("IS_TERMINATED", None),
                f"{_MDP}.rewards:lin_vel_z_l2": ("LIN_VEL_Z_L2", None), f"{_MDP}.rewards:ang_vel_xy_l2": ("ANG_VEL_XY_L2", None),
                f"{_MDP}.rewards:flat_orientation_l2": ("FLAT_ORIENTATION_L2", None),
                f"{_MDP}.rewards:joint_torques_l2": ("JOINT_TORQUES_L2", joints), f"{_MDP}.rewards:joint_vel_l1": ("JOINT_VEL_L1", joints),
                f"{_MDP}.rewards:joint_vel_l2": ("JOINT_VEL_L2", joints), f"{_MDP}.rewards:joint_acc_l2": ("JOINT_ACC_L2", joints),
                f"{_MDP}.rewards:joint_deviation_l1": ("JOINT_DEVIATION_L1", joints),
                f"{_MDP}.rewards:joint_pos_limits": ("JOINT_POS_LIMITS", joints),
                f"{_MDP}.rewards:applied_torque_limits": ("APPLIED_TORQUE_LIMITS", joints),
                f"{_MDP}.rewards:action_rate_l2": ("ACTION_RATE_L2", None), f"{_MDP}.rewards:action_l2": ("ACTION_L2", None),
            }
            if fn in table:
                opn, res = table[fn]
                rec["op"] = W_OPS[opn]
                if res:
                    res()
            elif fn == f"{_MDP}.rewards:is_terminated_term":
                keys = p.get("term_keys", ".*")
                ids, _ = resolve_matching_names(keys, [t.name for t in term_terms])
                rec.update(op=W_OPS["IS_TERMINATED_TERM"], ids_off=blob.ints(ids), nids=len(ids))
            elif fn == f"{_MDP}.rewards:base_height_l2" and p.get("sensor_cfg") is None:
                rec.update(op=W_OPS["BASE_HEIGHT_L2"], p0=f32(p["target_height"]))
            elif fn == f"{_MDP}.rewards:joint_vel_limits":
                joints()
                rec.update(op=W_OPS["JOINT_VEL_LIMITS"], p0=f32(p["soft_ratio"]))
            elif fn in (f"{_MDP}.rewards:undesired_contacts", f"{_MDP}.rewards:contact_forces"):
                bodies()
                rec.update(op=W_OPS["UNDESIRED_CONTACTS" if fn.endswith("undesired_contacts") else "CONTACT_FORCES"],
                           p0=f32(p["threshold"]))
            elif fn in (f"{_MDP}.rewards:track_lin_vel_xy_exp", f"{_MDP}.rewards:track_ang_vel_z_exp",
                        f"{_VEL}.rewards:track_lin_vel_xy_yaw_frame_exp", f"{_VEL}.rewards:track_ang_vel_z_world_exp"):
                opn = _short(fn)[1].upper()
                rec.update(op=W_OPS[opn], p0=f32(float(p["std"]) ** 2))  # python: std**2 in double, then fp32
            elif fn == f"{_VEL}.rewards:feet_air_time":
                bodies()
                rec.update(op=W_OPS["FEET_AIR_TIME"], p0=f32(p["threshold"]), p1=f32(step_dt + 1.0e-8))
            elif fn == f"{_VEL}.rewards:feet_air_time_positive_biped":
                bodies()
                rec.update(op=W_OPS["FEET_AIR_TIME_POSITIVE_BIPED"], p0=f32(p["threshold"]))
            elif fn == f"{_VEL}.rewards:feet_slide":
                bodies()
                ids2 = self.resolve_ids(p.get("asset_cfg"), "body")
                rec.update(op=W_OPS["FEET_SLIDE"], ids2_off=blob.ints(ids2), nids2=len(ids2))
            elif fn == f"{_CART}.rewards:joint_pos_target_l2":
                joints()
                rec.update(op=W_OPS["JOINT_POS_TARGET_L2"], p0=f32(p["target"]))
            elif fn == f"{_MDP}.rewards:body_lin_acc_l2":
                ids = self.resolve_ids(p.get("asset_cfg"), "body")
                rec.update(op=W_OPS["BODY_LIN_ACC_L2"], ids_off=blob.ints(ids), nids=len(ids))
            else:
                known = False
                rec.update(op=W_OPS["EXTERNAL"], aux0=n_ext_rew)
                n_ext_rew += 1
            term = Term(name, fn, rec["op"], p, external=None if known else tcfg["func"], weight=float(weight))
            rew_terms.append(term)
            rew_recs.append(_rec(**rec))  # zero-weight terms keep their record: the kernel skips them at run time, set_term_cfg can wake them

        # ---- observations (ObservationManager._prepare_terms, observation_manager.py:337-470): every group of the cfg, in cfg order;
        #      group g fills its own (N, D_g) tensor.  Record OUT = column inside the group, WEIGHT word = group index.
        obs_groups_cfg = cfg.get("observations") or {}
        group_names = [g_ for g_, v in obs_groups_cfg.items() if isinstance(v, dict)]
        if not group_names:
            raise ValueError("env cfg has no observation groups")
        if len(group_names) > MAX_OBS_GROUPS:
            raise NotImplementedError(f"{len(group_names)} observation groups; the fused path carries at most {MAX_OBS_GROUPS}")
        obs_recs: list[list[int]] = []
        groups: list[ObsGroup] = []
        n_ext_obs = 0
        mod_state = 0  # floats of modifier state per env
        ray_local = None
        scanner = scene.get("height_scanner")
        ray_dir = (0.0, 0.0, -1.0)
        ray_max = 1.0e6
        R_n = 0
        scan_primary = -1  # obs record index of the first height_scan term: later ones reuse its ray hits (one cast per ray and step)
        group_keys = ("concatenate_terms", "enable_corruption", "history_length", "flatten_history_dim")
        for gi, gname in enumerate(group_names):
            gcfg = obs_groups_cfg[gname]
            # concatenate_terms=False / flatten_history_dim=False change the SHAPE the manager hands out, not what is computed: the
            # kernel fills the same fused row, ObservationManager returns views of it (env.py)
            grp = ObsGroup(name=gname, enable_corruption=bool(gcfg.get("enable_corruption", False)), first_record=len(obs_recs),
                           concatenate=bool(gcfg.get("concatenate_terms", True)))
            D = 0
            for name, tcfg in gcfg.items():
                if name in group_keys or tcfg is None or not isinstance(tcfg, dict) or "func" not in tcfg:
                    continue
                fn = func_name(tcfg["func"])
                p = dict(tcfg.get("params") or {})
                # history (observation_manager.py:412-431): a group-level history_length overrides the terms'; the (N,H,d) window
                # is flattened oldest-first into H*d columns (flatten_history_dim); kept in the obs buffer itself by the kernel
                gh = gcfg.get("history_length")
                hist = int(gh if gh is not None else (tcfg.get("history_length") or 0))
                flat = gcfg.get("flatten_history_dim", True) if gh is not None else tcfg.get("flatten_history_dim", True)
                rec = dict(out=D, weight=int(gi))
                flags = 0
                known = True  # the term FUNCTION is one of the fused ops (else: evaluated by calling the Python term, IMX_O_EXTERNAL)
                # modifiers (observation_manager.py:310-312): modifier.py's five compile to a per-term program run by the kernel on the raw
                # value, whatever produced it.  A chain with a modifier from elsewhere is applied in Python, right after the (then
                # Python-evaluated) term function -- possible for function-style modifiers only (stateful classes need the manager)
                mod_prog, mod_slots, py_mods = [], 0, []
                if tcfg.get("modifiers"):
                    try:
                        mod_prog, mod_slots = compile_modifiers(tcfg["modifiers"])
                    except NotImplementedError:
                        for m in tcfg["modifiers"]:
                            m = m if isinstance(m, dict) else m.to_dict()
                            if _short(func_name(m["func"]))[1][:1].isupper():
                                raise NotImplementedError(
                                    f"observation term '{name}': class-based modifier {func_name(m['func'])} next to a modifier that is not one of "
                                    "isaaclab.utils.modifiers' five cannot run on the fused path")
                            py_mods.append((m["func"], dict(m.get("params") or {})))
                        known = False
                    last = _short(func_name(tcfg["modifiers"][-1]["func"] if isinstance(tcfg["modifiers"][-1], dict) else tcfg["modifiers"][-1].func))[1]
                    # the reference's Integrator returns its state tensor itself; a following in-place clip_/mul_ (no noise in
                    # between) writes into that state -- a reference quirk the fused path does not reproduce: refuse instead of differing
                    if last == "Integrator" and not (tcfg.get("noise") and grp.enable_corruption) and (tcfg.get("clip") is not None or tcfg.get("scale") is not None):
                        raise NotImplementedError(
                            f"observation term '{name}': an Integrator as last modifier followed by clip/scale without noise aliases the "
                            "integrator state in the reference (modifier.py:247-259, observation_manager.py:314-317); not supported")
                fixed = {f"{_MDP}.observations:base_pos_z": ("BASE_POS_Z", 1), f"{_MDP}.observations:base_lin_vel": ("BASE_LIN_VEL", 3),
                         f"{_MDP}.observations:base_ang_vel": ("BASE_ANG_VEL", 3),
                         f"{_MDP}.observations:projected_gravity": ("PROJECTED_GRAVITY", 3),
                         f"{_MDP}.observations:root_pos_w": ("ROOT_POS_W", 3), f"{_MDP}.observations:root_quat_w": ("ROOT_QUAT_W", 4),
                         f"{_MDP}.observations:root_lin_vel_w": ("ROOT_LIN_VEL_W", 3),
                         f"{_MDP}.observations:root_ang_vel_w": ("ROOT_ANG_VEL_W", 3)}
                jointy = {f"{_MDP}.observations:joint_pos": "JOINT_POS", f"{_MDP}.observations:joint_pos_rel": "JOINT_POS_REL",
                          f"{_MDP}.observations:joint_pos_limit_normalized": "JOINT_POS_LIMIT_NORMALIZED",
                          f"{_MDP}.observations:joint_vel": "JOINT_VEL", f"{_MDP}.observations:joint_vel_rel": "JOINT_VEL_REL"}
                dim = 0
                if not known:
                    pass
                elif fn in fixed:
                    opn, dim = fixed[fn]
                    rec["op"] = O_OPS[opn]
                    if opn == "ROOT_QUAT_W" and p.get("make_quat_unique"):
                        flags |= F_QUAT_UNIQUE
                elif fn in jointy:
                    ids = self.resolve_ids(p.get("asset_cfg"), "joint")
                    dim = len(ids)
                    rec.update(op=O_OPS[jointy[fn]], ids_off=blob.ints(ids), nids=dim)
                elif fn == f"{_MDP}.observations:last_action" and p.get("action_name") is None:
                    rec["op"], dim = O_OPS["LAST_ACTION"], A
                elif fn == f"{_MDP}.observations:generated_commands":
                    rec["op"], dim = O_OPS["GENERATED_COMMANDS"], 3
                elif fn == f"{_MDP}.observations:height_scan":
                    if scanner is None:
                        raise ValueError(f"Error while parsing '{name}:sensor_cfg'. The scene entity 'height_scanner' does not exist.")
                    if ray_local is None:
                        pc = scanner["pattern_cfg"]
                        if _short(func_name(pc["func"]))[1] != "grid_pattern":
                            raise NotImplementedError("only grid_pattern ray patterns are on the fused path")
                        starts, dirs = grid_pattern(pc["resolution"], pc["size"], tuple(pc.get("direction", (0.0, 0.0, -1.0))),
                                                    pc.get("ordering", "xy"))
                        off = scanner.get("offset") or {}
                        starts = starts + np.asarray(off.get("pos", (0.0, 0.0, 0.0)), np.float32)
                        d0 = _quat_apply_np(off.get("rot", (1.0, 0.0, 0.0, 0.0)), dirs[0])
                        ray_local, ray_dir, R_n = starts, tuple(float(x) for x in d0), len(starts)
                        ray_max = float(scanner.get("max_distance", 1.0e6))
                    rec.update(op=O_OPS["HEIGHT_SCAN"], p0=f32(p.get("offset", 0.5)))
                    dim = R_n
                    if scan_primary >= 0 and hist == 0:
                        flags |= F_SCAN_TWIN  # shares the rays of record `scan_primary` (AUX0)
                        rec["aux0"] = scan_primary
                else:
                    known = False
                if not known:
                    # the term function (and a foreign modifier chain) is evaluated in Python; the kernel still applies modifier.py's
                    # modifiers, uniform noise, clip and scale to the value.  Its width comes with the cfg.
                    rec = dict(out=D, weight=int(gi), op=O_OPS["EXTERNAL"], aux0=n_ext_obs)
                    flags = 0
                    dim = int(tcfg.get("_dim", 0))
                    if dim <= 0:
                        raise NotImplementedError(
                            f"observation term '{name}' ({fn}) is not on the fused path; give its width as cfg['_dim']")
                    n_ext_obs += dim
                    if hist > 0:
                        raise NotImplementedError(f"observation term '{name}': history on a term evaluated in Python is not supported")
                noise = tcfg.get("noise")
                if noise:  # the reference's three noise functions with scalar parameters run in the kernel; anything else must not be dropped silently
                    nfn = _short(func_name(noise["func"]))[1]
                    num = lambda *ks: all(isinstance(noise.get(k_), (int, float)) for k_ in ks)  # noqa: E731
                    opbit = {"add": F_NOISE_ADD, "scale": F_NOISE_SCALE, "abs": F_NOISE_ABS}.get(noise.get("operation", "add"))
                    if opbit is None:
                        raise ValueError(f"Unknown operation in noise: {noise.get('operation')}")  # noise_model.py:38,68,94
                    if nfn == "uniform_noise" and num("n_min", "n_max"):
                        flags |= opbit
                        rec.update(noise_lo=f32(noise["n_min"]), noise_hi=f32(noise["n_max"]))
                    elif nfn == "constant_noise" and num("bias"):  # u * (b - b) + b == b for every u: the uniform path, bit-identical
                        flags |= opbit
                        rec.update(noise_lo=f32(noise["bias"]), noise_hi=f32(noise["bias"]))
                    elif nfn == "gaussian_noise" and num("mean", "std"):
                        flags |= opbit | F_NOISE_GAUSS
                        rec.update(noise_lo=f32(noise["mean"]), noise_hi=f32(noise["std"]))
                    elif grp.enable_corruption:
                        raise NotImplementedError(f"observation term '{name}': noise model {func_name(noise['func'])} is not on the fused path "
                                                  "(uniform_noise, constant_noise and gaussian_noise with scalar parameters are)")
                if tcfg.get("clip") is not None:
                    flags |= F_CLIP
                    rec.update(clip_lo=f32(tcfg["clip"][0]), clip_hi=f32(tcfg["clip"][1]))
                if tcfg.get("scale") is not None:
                    if not isinstance(tcfg["scale"], (int, float)):
                        raise NotImplementedError(f"observation term '{name}': only a scalar `scale` is on the fused path")
                    flags |= F_SCALE
                    rec["scale"] = f32(tcfg["scale"])
                if rec["op"] == O_OPS["HEIGHT_SCAN"] and scan_primary < 0:
                    scan_primary = len(obs_recs)
                if mod_prog:
                    flags |= F_MODIFIERS
                    rec.update(ids2_off=blob.ints(mod_prog), nids2=len(mod_prog), p1=int(mod_state))
                    mod_state += mod_slots * dim
                width = max(hist, 1) * dim
                rec.update(dim=dim, flags=flags, aux1=hist)
                obs_recs.append(_rec(**rec))
                grp.terms.append(Term(name, fn, rec["op"], p, external=None if known else tcfg["func"], dim=width, py_modifiers=py_mods))
                grp.term_dims.append((hist, dim) if hist > 0 and not flat else (width,))
                grp.term_widths.append(width)
                D += width
            if grp.concatenate and len({len(d) for d in grp.term_dims}) > 1:  # observation_manager.py:89-99
                raise RuntimeError(f"Unable to concatenate observation terms in group '{gname}'. The shapes of the terms are: {grp.term_dims}."
                                   " Please ensure that the shapes are compatible for concatenation. Otherwise, set 'concatenate_terms' to False"
                                   " in the group configuration.")
            grp.dim = D
            grp.num_records = len(obs_recs) - grp.first_record
            groups.append(grp)
        D = sum(g_.dim for g_ in groups)
        obs_terms = groups[0].terms
        obs_dims = groups[0].term_dims
        corruption = any(g_.enable_corruption for g_ in groups)

        # ---- height scanner as a SensorBase: update_period gating and drift (sensor_base.py:196-205,287-297; ray_caster.py:107-114)
        scan_period = float((scanner or {}).get("update_period", 0.0) or 0.0)
        drift = tuple((scanner or {}).get("drift_range", (0.0, 0.0)) or (0.0, 0.0))
        scan_stateful = bool(R_n > 0 and (scan_period > 0.0 or drift[0] != 0.0 or drift[1] != 0.0))

        # ---- assemble
        # the step kernel stages words [HEADER_WORDS, end of the reward table) in LDS (id lists + termination and reward records):
        # keep that range small -- the ray table and the observation / action records come after it
        group_off = blob.ints([x for g_ in groups for x in (g_.dim, int(g_.enable_corruption), g_.first_record, g_.num_records)])
        term_off = blob.table(term_recs)
        rew_off = blob.table(rew_recs)
        ray_off = blob.floats(ray_local.reshape(-1)) if ray_local is not None else 0
        obs_off = blob.table(obs_recs)
        act_off = blob.table(act_recs)
        w = blob.w
        hdr = {
            "MAGIC": MAGIC, "VERSION": PLAN_VERSION, "J": J, "B": B, "H": Hh, "A": A, "D": D, "R": R_n,
            "NTERM": len(term_recs), "NREW": len(rew_recs), "NOBS": len(obs_recs), "NACT": len(act_recs),
            "MAX_EP_LEN": max_len, "TERM_OFF": term_off, "REW_OFF": rew_off, "OBS_OFF": obs_off, "ACT_OFF": act_off,
            "TOTAL_WORDS": len(w), "NB": B, "NREW_ALL": len(rew_terms), "RAY_OFF": ray_off,
            "NEXT_REW": n_ext_rew, "NEXT_TERM": n_ext_term, "NEXT_OBS": n_ext_obs,
            "RAY_YAW_ONLY": 1 if (scanner and scanner.get("attach_yaw_only")) else 0, "CMD_DIM": 3,
            "MOD_STATE": mod_state, "NGROUPS": len(groups), "GROUP_OFF": group_off, "SCAN_SUBSTEPS": int(cfg["decimation"]),
            "SCAN_STATEFUL": int(scan_stateful),
        }
        for k, v in hdr.items():
            w[H[k]] = int(v)
        for k, v in {"STEP_DT": f32(step_dt), "GRAV_X": gdir[0], "GRAV_Y": gdir[1], "GRAV_Z": gdir[2],
                     "RAYDIR_X": ray_dir[0], "RAYDIR_Y": ray_dir[1], "RAYDIR_Z": ray_dir[2], "RAY_MAXDIST": ray_max,
                     "MAX_EP_LEN_S": f32(max_len_s), "SCAN_PERIOD": f32(scan_period), "SCAN_DT": f32(cfg["sim"]["dt"]),
                     "SCAN_DRIFT_LO": f32(drift[0]), "SCAN_DRIFT_HI": f32(drift[1])}.items():
            w[H[k]] = _f2w(float(v))
        arr = np.asarray(w, dtype=np.int64)
        arr = np.where(arr >= 2 ** 31, arr - 2 ** 32, arr).astype(np.int32)
        return Plan(blob=arr, robot=robot, num_joints=J, num_bodies=B, history=Hh, action_dim=A, obs_dim=groups[0].dim, num_rays=R_n,
                    obs_groups=groups, obs_dim_total=D, scan_stateful=scan_stateful, scan_drift_range=(float(drift[0]), float(drift[1])),
                    cmd_dim=3, step_dt=step_dt, max_episode_length=max_len, max_episode_length_s=max_len_s,
                    is_finite_horizon=bool(cfg.get("is_finite_horizon", False)), reward_terms=rew_terms,
                    termination_terms=term_terms, obs_terms=obs_terms, obs_term_dims=obs_dims,
                    action_terms=action_terms, enable_corruption=corruption, ray_starts_local=ray_local,
                    ray_direction=ray_dir, ray_max_distance=ray_max, scanner_cfg=scanner, n_ext_rew=n_ext_rew,
                    n_ext_term=n_ext_term, n_ext_obs=n_ext_obs, gravity_dir=tuple(float(x) for x in gdir),
                    mod_state_dim=mod_state)


def compile_plan(env_cfg: Any, robot: RobotSpec) -> Plan:
    return PlanCompiler(env_cfg, robot).compile()
