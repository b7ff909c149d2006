"""Articulation name tables and defaults for the configs in BASELINE.json.

The reference obtains joint/body names and default joint positions from PhysX after loading the robot USD
(``Articulation._initialize_impl``); neither PhysX nor the USD files exist here, so the *names* ship as data.
Defaults follow the asset cfgs (reference ``source/isaaclab_assets/isaaclab_assets/robots/anymal.py:112-121``,
``unitree.py:290-307`` (G1), ``cartpole.py``).  The G1 joint order is a synthetic breadth-first order of the 37
joint names the G1 cfg's regexes refer to (the true PhysX order is not recoverable offline); term semantics do
not depend on it because every index list is resolved by name through :func:`resolve_matching_names`.
"""

from __future__ import annotations

import dataclasses
import math
import re
from collections.abc import Sequence


def resolve_matching_names(keys, list_of_strings: Sequence[str], preserve_order: bool = False):
    """Regex name resolution with the reference's semantics (``isaaclab/utils/string.py:178-271``).

    Each target string may match at most one key (else ``ValueError``); every key must match something (else
    ``ValueError``).  Result order follows ``list_of_strings`` unless ``preserve_order`` (then key order).
    """
    if isinstance(keys, str):
        keys = [keys]
    hits: list[tuple[int, int]] = []  # (key index, target index)
    matched_by: list[str | None] = [None] * len(list_of_strings)
    key_hits = [0] * len(keys)
    for ti, s in enumerate(list_of_strings):
        for ki, k in enumerate(keys):
            if re.fullmatch(k, s):
                if matched_by[ti]:
                    raise ValueError(f"Multiple matches for '{s}': '{matched_by[ti]}' and '{k}'!")
                matched_by[ti] = k
                hits.append((ki, ti))
                key_hits[ki] += 1
    if not all(key_hits):
        missing = [k for k, c in zip(keys, key_hits) if not c]
        raise ValueError(
            f"Not all regular expressions are matched! Unmatched: {missing}. Available strings: {list(list_of_strings)}"
        )
    if preserve_order:
        hits.sort(key=lambda kt: kt[0])  # stable: target order inside one key
    idx = [ti for _, ti in hits]
    return idx, [list_of_strings[i] for i in idx]


def resolve_matching_names_values(data: dict, list_of_strings: Sequence[str]):
    """``{regex: value}`` -> (indices, names, values) in target order (``isaaclab/utils/string.py:274-360``)."""
    idx, names, vals = [], [], []
    matched_by: list[str | None] = [None] * len(list_of_strings)
    key_hits = {k: 0 for k in data}
    for ti, s in enumerate(list_of_strings):
        for k, v in data.items():
            if re.fullmatch(k, s):
                if matched_by[ti]:
                    raise ValueError(f"Multiple matches for '{s}': '{matched_by[ti]}' and '{k}'!")
                matched_by[ti] = k
                idx.append(ti)
                names.append(s)
                vals.append(v)
                key_hits[k] += 1
    if not all(key_hits.values()):
        raise ValueError(f"Not all regular expressions are matched! {key_hits}")
    return idx, names, vals


@dataclasses.dataclass
class RobotSpec:
    name: str
    joint_names: list[str]
    body_names: list[str]
    default_joint_pos: dict  # regex -> value (InitialStateCfg.joint_pos)
    default_root_height: float
    soft_joint_pos_limit_factor: float = 1.0
    joint_pos_limits: tuple[float, float] = (-2.0 * math.pi, 2.0 * math.pi)
    joint_vel_limit: float = 100.0

    @property
    def num_joints(self) -> int:
        return len(self.joint_names)

    @property
    def num_bodies(self) -> int:
        return len(self.body_names)

    def default_joint_pos_list(self) -> list[float]:
        out = [0.0] * self.num_joints
        idx, _, vals = resolve_matching_names_values(self.default_joint_pos, self.joint_names)
        for i, v in zip(idx, vals):
            out[i] = float(v)
        return out

    def soft_joint_pos_limits(self) -> list[tuple[float, float]]:
        # Articulation._process_cfg: mean +- 0.5 * range * factor  (reference articulation.py soft limits)
        lo, hi = self.joint_pos_limits
        mean, rng = 0.5 * (lo + hi), hi - lo
        f = self.soft_joint_pos_limit_factor
        return [(mean - 0.5 * rng * f, mean + 0.5 * rng * f)] * self.num_joints


_LEGS = ("LF", "LH", "RF", "RH")

ANYMAL_C = RobotSpec(
    name="anymal_c",
    # PhysX breadth-first order of anymal_c.usd
    joint_names=[f"{leg}_{j}" for j in ("HAA", "HFE", "KFE") for leg in _LEGS],
    body_names=["base"] + [f"{leg}_{b}" for b in ("HIP", "THIGH", "SHANK", "FOOT") for leg in _LEGS],
    default_joint_pos={".*HAA": 0.0, ".*F_HFE": 0.4, ".*H_HFE": -0.4, ".*F_KFE": -0.8, ".*H_KFE": 0.8},
    default_root_height=0.6,
    soft_joint_pos_limit_factor=0.95,
    joint_vel_limit=7.5,
)

_G1_JOINTS = [
    "left_hip_pitch_joint", "right_hip_pitch_joint", "torso_joint",
    "left_hip_roll_joint", "right_hip_roll_joint", "left_shoulder_pitch_joint", "right_shoulder_pitch_joint",
    "left_hip_yaw_joint", "right_hip_yaw_joint", "left_shoulder_roll_joint", "right_shoulder_roll_joint",
    "left_knee_joint", "right_knee_joint", "left_shoulder_yaw_joint", "right_shoulder_yaw_joint",
    "left_ankle_pitch_joint", "right_ankle_pitch_joint", "left_elbow_pitch_joint", "right_elbow_pitch_joint",
    "left_ankle_roll_joint", "right_ankle_roll_joint", "left_elbow_roll_joint", "right_elbow_roll_joint",
    "left_five_joint", "left_three_joint", "left_zero_joint", "right_five_joint", "right_three_joint",
    "right_zero_joint", "left_six_joint", "left_four_joint", "left_one_joint", "right_six_joint",
    "right_four_joint", "right_one_joint", "left_two_joint", "right_two_joint",
]

G1 = RobotSpec(
    name="g1",
    joint_names=_G1_JOINTS,
    body_names=["pelvis"] + [j.replace("_joint", "_link") for j in _G1_JOINTS],
    default_joint_pos={
        ".*_hip_pitch_joint": -0.20, ".*_knee_joint": 0.42, ".*_ankle_pitch_joint": -0.23,
        ".*_elbow_pitch_joint": 0.87, "left_shoulder_roll_joint": 0.16, "left_shoulder_pitch_joint": 0.35,
        "right_shoulder_roll_joint": -0.16, "right_shoulder_pitch_joint": 0.35, "left_one_joint": 1.0,
        "right_one_joint": -1.0, "left_two_joint": 0.52, "right_two_joint": -0.52,
    },
    default_root_height=0.74,
    soft_joint_pos_limit_factor=0.9,
)

CARTPOLE = RobotSpec(
    name="cartpole",
    joint_names=["slider_to_cart", "cart_to_pole"],
    body_names=["rail", "cart", "pole"],
    default_joint_pos={"slider_to_cart": 0.0, "cart_to_pole": 0.0},
    default_root_height=2.0,
    joint_pos_limits=(-4.0, 4.0),
)

ROBOTS = {r.name: r for r in (ANYMAL_C, G1, CARTPOLE)}
