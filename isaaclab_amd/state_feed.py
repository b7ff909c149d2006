"""Synthetic/recorded articulation-state feed: stands in for PhysX + ``InteractiveScene`` (out of scope).

The reference reads per-step tensors from ``ArticulationData`` (``isaaclab/assets/articulation/articulation_data.py``),
``ContactSensorData`` (``isaaclab/sensors/contact_sensor/contact_sensor_data.py``) and the command manager.  Here those
tensors come from a :class:`StateFeed`: ``S`` pre-generated snapshots resident in HBM, cycled one per env-step, so the
reference path (oracle) and the HIP path are fed *identical* tensors of shape ``(num_envs, dof|bodies)``.
Distributions follow SURVEY.md section 8(d); generation is on CPU with ``torch.Generator().manual_seed(seed)``.
"""

from __future__ import annotations

import math

import torch

from .robots import RobotSpec

# tensors that change every physics step (one copy per snapshot)
DYNAMIC = (
    "root_pos_w", "root_quat_w", "root_lin_vel_w", "root_ang_vel_w",
    "joint_pos", "joint_vel", "joint_acc", "applied_torque", "computed_torque",
    "body_lin_vel_w", "command", "net_forces_w_history",
    "last_air_time", "current_air_time", "current_contact_time", "last_contact_time",
)
# further per-step tensors only a few optional terms read (body_lin_acc_l2, command_resample): drawn from their own generator so that
# the tensors above are unchanged by their presence; recorded fixtures carry them only when their cfg needs them
EXTRA = ("body_lin_acc_w", "command_time_left", "command_counter")
# tensors fixed for the lifetime of the scene
STATIC = ("default_joint_pos", "default_joint_vel", "soft_joint_pos_limits", "soft_joint_vel_limits", "env_origins")


def _quat_from_euler(roll, pitch, yaw):
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    return torch.stack(
        [cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp],
        dim=-1,
    )


def contact_body_groups(robot: RobotSpec) -> dict[str, list[int]]:
    """Which bodies play the 'feet' / 'thigh' / 'base' roles for the synthetic contact distribution."""
    names = robot.body_names
    feet = [i for i, n in enumerate(names) if n.endswith("FOOT") or n.endswith("_ankle_roll_link")]
    thigh = [i for i, n in enumerate(names) if n.endswith("THIGH") or n.endswith("_knee_link")]
    base = [i for i, n in enumerate(names) if n in ("base", "torso_link")]
    return {"feet": feet, "thigh": thigh, "base": base}


def generate_snapshot(robot: RobotSpec, num_envs: int, gen: torch.Generator, history: int = 3,
                      env_spacing: float = 2.5, extent_xy: tuple[float, float] | None = None) -> dict[str, torch.Tensor]:
    """One synthetic post-physics state (CPU tensors).  ``extent_xy``: half-size of the terrain the env origins
    are folded into (so the height-scanner footprint stays on the mesh)."""
    N, J, B = num_envs, robot.num_joints, robot.num_bodies
    f32 = torch.float32

    def U(lo, hi, *shape):
        return torch.rand(*shape, generator=gen, dtype=f32) * (hi - lo) + lo

    def Nrm(std, *shape):
        return torch.randn(*shape, generator=gen, dtype=f32) * std

    out: dict[str, torch.Tensor] = {}
    # env-origin grid (InteractiveScene grid cloner: square grid, spacing 2.5 m, centred)
    rows = int(math.ceil(math.sqrt(N)))
    ii = torch.arange(N)
    ox = (ii // rows).to(f32) * env_spacing
    oy = (ii % rows).to(f32) * env_spacing
    ox -= ox.mean()
    oy -= oy.mean()
    if extent_xy is not None:  # fold into the terrain extent
        ex, ey = extent_xy
        ox = torch.remainder(ox + ex, 2 * ex) - ex
        oy = torch.remainder(oy + ey, 2 * ey) - ey
    origins = torch.stack([ox, oy, torch.zeros(N)], dim=-1)
    out["env_origins"] = origins
    pos = origins.clone()
    # never lattice-aligned: irrational-ish offsets
    pos[:, 0] += U(-0.5, 0.5, N) + 0.0137
    pos[:, 1] += U(-0.5, 0.5, N) + 0.0071
    pos[:, 2] = robot.default_root_height + Nrm(0.03, N)
    out["root_pos_w"] = pos
    out["root_quat_w"] = _quat_from_euler(Nrm(0.15, N), Nrm(0.15, N), U(-math.pi, math.pi, N))
    out["root_lin_vel_w"] = Nrm(0.5, N, 3)
    out["root_ang_vel_w"] = Nrm(0.5, N, 3)

    default = torch.tensor(robot.default_joint_pos_list(), dtype=f32).repeat(N, 1)
    out["default_joint_pos"] = default
    out["default_joint_vel"] = torch.zeros(N, J, dtype=f32)
    out["joint_pos"] = default + U(-0.5, 0.5, N, J)
    out["joint_vel"] = Nrm(1.0, N, J)
    out["joint_acc"] = Nrm(20.0, N, J)
    out["computed_torque"] = Nrm(20.0, N, J)
    # actuator clipping: ~10 % of the efforts saturate (applied != computed)
    out["applied_torque"] = out["computed_torque"].clamp(-32.0, 32.0)
    lim = torch.tensor(robot.soft_joint_pos_limits(), dtype=f32)  # (J,2)
    # tighter synthetic limits so that joint_pos_limits / limit terms are exercised
    lim = torch.stack([default[0] - 0.45, default[0] + 0.45], dim=-1) if robot.name != "cartpole" else lim
    out["soft_joint_pos_limits"] = lim.unsqueeze(0).repeat(N, 1, 1).contiguous()
    out["soft_joint_vel_limits"] = torch.full((N, J), 2.0, dtype=f32)
    out["body_lin_vel_w"] = Nrm(0.5, N, B, 3)

    cmd = U(-1.0, 1.0, N, 3)
    cmd[torch.rand(N, generator=gen) < 0.02] = 0.0
    out["command"] = cmd

    groups = contact_body_groups(robot)
    forces = torch.zeros(N, history, B, 3, dtype=f32)

    def fill(ids, prob, std):
        if not ids:
            return
        k = len(ids)
        on = (torch.rand(N, history, k, generator=gen) < prob).to(f32)
        direction = torch.randn(N, history, k, 3, generator=gen, dtype=f32)
        direction = direction / direction.norm(dim=-1, keepdim=True).clamp_min(1e-6)
        mag = Nrm(std, N, history, k).abs()
        forces[:, :, ids, :] = direction * (mag * on).unsqueeze(-1)

    fill(groups["feet"], 0.5, 60.0)
    fill(groups["thigh"], 0.05, 30.0)
    fill(groups["base"], 0.01, 30.0)
    out["net_forces_w_history"] = forces
    out["last_air_time"] = U(0.0, 0.8, N, B)
    out["current_air_time"] = U(0.0, 0.8, N, B)
    out["last_contact_time"] = U(0.0, 0.8, N, B)
    sel = torch.rand(N, B, generator=gen)
    cct = torch.where(sel < 0.34, torch.zeros(N, B), torch.where(sel < 0.67, U(0.0, 0.04, N, B), U(0.04, 1.0, N, B)))
    out["current_contact_time"] = cct
    # a body in contact has zero air time (ContactSensor bookkeeping, contact_sensor.py:352-379)
    out["current_air_time"] = torch.where(cct > 0, torch.zeros(N, B), out["current_air_time"])
    return out


def generate_extras(robot: RobotSpec, num_envs: int, gen: torch.Generator) -> dict[str, torch.Tensor]:
    """``ArticulationData.body_lin_acc_w`` and the command term's ``time_left`` / ``command_counter`` (CommandTerm,
    isaaclab/managers/command_manager.py:55-62): N(0, 5) accelerations; time_left ~ U(0, 0.1) s and counters in {0,1,2} so that
    ``command_resample`` fires on a few envs."""
    N, B = num_envs, robot.num_bodies
    return {"body_lin_acc_w": torch.randn(N, B, 3, generator=gen) * 5.0,
            "command_time_left": torch.rand(N, generator=gen) * 0.1,
            "command_counter": torch.randint(0, 3, (N,), generator=gen, dtype=torch.int64)}


class StateFeed:
    """``S`` snapshots of the post-physics state held on ``device``; ``advance()`` moves to the next one.

    ``feed[name]`` returns the current tensor for ``name`` (a view into the snapshot stack; static tensors are
    shared).  ``feed.index`` identifies the current snapshot so callers can cache per-snapshot pointer structs.
    """

    def __init__(self, robot: RobotSpec, num_envs: int, device: str | torch.device = "cpu", seed: int = 42,
                 num_snapshots: int = 4, history: int = 3, extent_xy: tuple[float, float] | None = None,
                 gravity=(0.0, 0.0, -9.81)):
        self.robot = robot
        self.num_envs = num_envs
        self.device = torch.device(device)
        self.num_snapshots = num_snapshots
        self.history = history
        gen = torch.Generator().manual_seed(seed)
        snaps = [generate_snapshot(robot, num_envs, gen, history, extent_xy=extent_xy) for _ in range(num_snapshots)]
        self._stack: dict[str, torch.Tensor] = {}
        gen_x = torch.Generator().manual_seed(seed + 0x5EED)
        for sn in snaps:
            sn.update(generate_extras(robot, num_envs, gen_x))
        for name in DYNAMIC + EXTRA:
            self._stack[name] = torch.stack([s[name] for s in snaps], dim=0).to(self.device).contiguous()
        # every snapshot keeps the same origins/defaults; root xy of later snapshots re-uses snapshot-0 origins
        self._static = {name: snaps[0][name].to(self.device).contiguous() for name in STATIC}
        for k in range(1, num_snapshots):
            delta = (snaps[k]["root_pos_w"] - snaps[k]["env_origins"]).to(self.device)
            self._stack["root_pos_w"][k] = self._static["env_origins"] + delta
        g = torch.tensor(gravity, dtype=torch.float32)
        self.gravity_dir = (g / g.norm().clamp_min(1e-9)).tolist()
        self.index = 0

    @classmethod
    def from_tensors(cls, robot: RobotSpec, snapshots: list[dict[str, torch.Tensor]], device="cpu",
                     gravity_dir=(0.0, 0.0, -1.0)) -> "StateFeed":
        """Recorded feed: ``snapshots[k][name]`` for every DYNAMIC name; STATIC names taken from ``snapshots[0]``."""
        self = cls.__new__(cls)
        self.robot = robot
        self.device = torch.device(device)
        self.num_snapshots = len(snapshots)
        self.num_envs = snapshots[0]["root_pos_w"].shape[0]
        self.history = snapshots[0]["net_forces_w_history"].shape[1]
        self._stack = {
            n: torch.stack([torch.as_tensor(s[n]) for s in snapshots], 0).to(self.device).contiguous()
            for n in DYNAMIC + EXTRA if n in snapshots[0]
        }
        self._static = {n: torch.as_tensor(snapshots[0][n]).to(self.device).contiguous() for n in STATIC}
        self.gravity_dir = [float(x) for x in gravity_dir]
        self.index = 0
        return self

    def physx(self, name: str) -> torch.Tensor:
        """The current snapshot in the layout PhysX tensor views deliver (articulation_data.py:365-380): ``root_transforms`` (N,7) = position +
        quaternion XYZW, ``root_velocities`` (N,6) = linear + angular.  Built once per feed, resident like the snapshots."""
        if getattr(self, "_physx", None) is None:
            q = self._stack["root_quat_w"]
            self._physx = {"root_transforms": torch.cat([self._stack["root_pos_w"], q[..., 1:4], q[..., 0:1]], dim=-1).contiguous(),
                           "root_velocities": torch.cat([self._stack["root_lin_vel_w"], self._stack["root_ang_vel_w"]], dim=-1).contiguous()}
        return self._physx[name][self.index]

    def advance(self) -> int:
        self.index = (self.index + 1) % self.num_snapshots
        return self.index

    def seek(self, index: int):
        self.index = index % self.num_snapshots

    def __getitem__(self, name: str) -> torch.Tensor:
        if name in self._static:
            return self._static[name]
        return self._stack[name][self.index]

    def snapshot(self, index: int | None = None) -> dict[str, torch.Tensor]:
        k = self.index if index is None else index
        d = {n: t[k] for n, t in self._stack.items()}
        d.update(self._static)
        return d

    def names(self):
        return tuple(self._stack) + tuple(STATIC)
