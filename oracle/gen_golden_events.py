"""TEST INFRASTRUCTURE (build container only): tests/golden/events.npz from the REAL reference reset / interval event
terms (isaaclab/envs/mdp/events.py: reset_root_state_uniform, reset_joints_by_scale, reset_joints_by_offset,
push_by_setting_velocity) and the terrain curriculum (isaaclab_tasks/.../velocity/mdp/curriculums.py::terrain_levels_vel ->
TerrainImporter.update_env_origins), run against a fake asset that records what would be written to the simulator.

Every torch.rand / torch.randint_like draw of the terms is re-drawn after re-seeding (same shapes, same order) and stored,
scattered to env rows, so the HIP path can be fed the same samples ("RNG draws are the parity
obstacle").  See oracle/gen_golden.py for the import stub.
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

from isaaclab.envs.mdp import events as ref_events  # noqa: E402
from isaaclab.managers import SceneEntityCfg  # noqa: E402
from isaaclab.terrains.terrain_importer import TerrainImporter  # noqa: E402
from isaaclab_tasks.manager_based.locomotion.velocity.mdp.curriculums import terrain_levels_vel  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


class FakeAsset:
    device = "cpu"

    def __init__(self, N, J, g):
        q = torch.randn(N, 4, generator=g)
        q = q / q.norm(dim=-1, keepdim=True)
        drs = torch.cat([torch.randn(N, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, 0.6]), q,
                         0.05 * torch.randn(N, 6, generator=g)], dim=-1)
        lim_c = torch.randn(N, J, generator=g) * 0.3
        self.data = types.SimpleNamespace(
            default_root_state=drs,
            default_joint_pos=torch.randn(N, J, generator=g) * 0.6,
            default_joint_vel=torch.randn(N, J, generator=g) * 0.2,
            soft_joint_pos_limits=torch.stack([lim_c - 0.5, lim_c + 0.5], dim=-1),
            soft_joint_vel_limits=torch.rand(N, J, generator=g) * 0.2 + 0.05,
            root_vel_w=torch.randn(N, 6, generator=g),
            root_pos_w=torch.randn(N, 3, generator=g) * 3.0,
        )
        self.writes = {}

    def write_root_pose_to_sim(self, pose, env_ids=None):
        self.writes["root_pose"] = (pose.clone(), env_ids.clone())

    def write_root_velocity_to_sim(self, vel, env_ids=None):
        self.writes["root_vel"] = (vel.clone(), env_ids.clone())

    num_bodies = 17

    def set_external_force_and_torque(self, forces, torques, env_ids=None, body_ids=None):
        self.writes["ext"] = (forces.clone(), torques.clone(), env_ids.clone(), body_ids)

    def write_joint_state_to_sim(self, pos, vel, env_ids=None):
        self.writes["joint_pos"] = (pos.clone(), env_ids.clone())
        self.writes["joint_vel"] = (vel.clone(), env_ids.clone())


class FakeScene(dict):
    pass


def scatter(N, ids, rows):
    out = torch.zeros((N,) + tuple(rows.shape[1:]), dtype=rows.dtype)
    out[ids] = rows
    return out


def main():
    N, J = 96, 12
    g = torch.Generator().manual_seed(23)
    asset = FakeAsset(N, J, g)
    scene = FakeScene(robot=asset)
    scene.env_origins = torch.randn(N, 3, generator=g) * 4.0
    env = types.SimpleNamespace(scene=scene)
    mask = torch.rand(N, generator=g) < 0.4
    mask[0], mask[N - 1] = True, False
    ids = mask.nonzero(as_tuple=False).squeeze(-1)
    k = len(ids)
    cfg = SceneEntityCfg("robot")
    rec = {"mask": mask.numpy(), "env_origins": scene.env_origins.numpy().copy()}
    for name in ("default_root_state", "default_joint_pos", "default_joint_vel", "soft_joint_pos_limits", "soft_joint_vel_limits",
                 "root_vel_w", "root_pos_w"):
        rec[name] = getattr(asset.data, name).numpy().copy()

    # ---- reset_root_state_uniform (velocity_env_cfg.py:187-202 ranges, plus non-zero roll/pitch/z to exercise every axis)
    pose_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (0.0, 0.1), "roll": (-0.3, 0.3), "pitch": (-0.2, 0.2), "yaw": (-3.14, 3.14)}
    vel_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5), "z": (-0.5, 0.5), "roll": (-0.5, 0.5), "pitch": (-0.5, 0.5), "yaw": (-0.5, 0.5)}
    torch.manual_seed(101)
    ref_events.reset_root_state_uniform(env, ids, pose_range, vel_range, cfg)
    torch.manual_seed(101)
    u_pose, u_vel = torch.rand(k, 6), torch.rand(k, 6)
    rec["root/u_pose"], rec["root/u_vel"] = scatter(N, ids, u_pose).numpy(), scatter(N, ids, u_vel).numpy()
    rec["root/pose_out"] = scatter(N, ids, asset.writes["root_pose"][0]).numpy()
    rec["root/vel_out"] = scatter(N, ids, asset.writes["root_vel"][0]).numpy()
    assert torch.equal(asset.writes["root_pose"][1], ids)

    # ---- reset_joints_by_scale (velocity_env_cfg.py:204-211: position_range (0.5, 1.5), velocity_range (0, 0)) and by_offset
    for tag, fn, prange, vrange, seed in (("scale", ref_events.reset_joints_by_scale, (0.5, 1.5), (0.0, 0.0), 102),
                                          ("scale2", ref_events.reset_joints_by_scale, (0.2, 2.5), (-1.5, 1.5), 103),
                                          ("offset", ref_events.reset_joints_by_offset, (-0.8, 0.8), (-0.3, 0.3), 104)):
        torch.manual_seed(seed)
        fn(env, ids, prange, vrange, cfg)
        torch.manual_seed(seed)
        u_p, u_v = torch.rand(k, J), torch.rand(k, J)
        rec[f"joints_{tag}/u_pos"], rec[f"joints_{tag}/u_vel"] = scatter(N, ids, u_p).numpy(), scatter(N, ids, u_v).numpy()
        rec[f"joints_{tag}/pos_out"] = scatter(N, ids, asset.writes["joint_pos"][0]).numpy()
        rec[f"joints_{tag}/vel_out"] = scatter(N, ids, asset.writes["joint_vel"][0]).numpy()
        rec[f"joints_{tag}/ranges"] = np.array([*prange, *vrange], dtype=np.float32)

    # ---- push_by_setting_velocity (velocity_env_cfg.py:214-219: x, y in (-0.5, 0.5))
    push_range = {"x": (-0.5, 0.5), "y": (-0.5, 0.5)}
    torch.manual_seed(105)
    ref_events.push_by_setting_velocity(env, ids, push_range, cfg)
    torch.manual_seed(105)
    rec["push/u"] = scatter(N, ids, torch.rand(k, 6)).numpy()
    rec["push/vel_out"] = scatter(N, ids, asset.writes["root_vel"][0]).numpy()

    # ---- apply_external_force_torque (velocity_env_cfg.py:177-185 uses zero ranges on the base; non-zero here, 3 of 17 bodies)
    body_cfg = SceneEntityCfg("robot")
    body_cfg.body_ids = [0, 5, 11]
    torch.manual_seed(107)
    ref_events.apply_external_force_torque(env, ids, (-2.0, 3.0), (-0.5, 0.7), body_cfg)
    torch.manual_seed(107)
    rec["ext/u_force"] = scatter(N, ids, torch.rand(k, 3, 3)).numpy()
    rec["ext/u_torque"] = scatter(N, ids, torch.rand(k, 3, 3)).numpy()
    fo, to, eids, bids = asset.writes["ext"]
    assert torch.equal(eids, ids) and bids == [0, 5, 11]
    rec["ext/forces"], rec["ext/torques"] = scatter(N, ids, fo).numpy(), scatter(N, ids, to).numpy()
    rec["ext/body_ids"] = np.array([0, 5, 11], dtype=np.int32)
    rec["ext/ranges"] = np.array([-2.0, 3.0, -0.5, 0.7], dtype=np.float32)

    # ---- terrain_levels_vel + TerrainImporter.update_env_origins
    R, C = 10, 20
    ti = TerrainImporter.__new__(TerrainImporter)
    ti.cfg = types.SimpleNamespace(terrain_generator=types.SimpleNamespace(size=(8.0, 8.0)))
    ti.terrain_origins = torch.randn(R, C, 3, generator=g) * 10.0
    ti.max_terrain_level = R
    ti.terrain_levels = torch.randint(0, R, (N,), generator=g)
    ti.terrain_levels[ids[:4]] = R - 1  # some at the top level: move_up sends them to a random level
    ti.terrain_levels[ids[4:8]] = 0     # some at the bottom: move_down clips at zero
    ti.terrain_types = torch.randint(0, C, (N,), generator=g)
    ti.env_origins = ti.terrain_origins[ti.terrain_levels, ti.terrain_types].clone()
    scene.terrain = ti
    scene.env_origins = ti.env_origins
    command = torch.rand(N, 3, generator=g) * 2.0 - 1.0
    env.command_manager = types.SimpleNamespace(get_command=lambda name: command)
    env.max_episode_length_s = 20.0
    # walked distances around both thresholds (4 m; 0.5 * |cmd_xy| * 20 s)
    walked = torch.rand(N, generator=g) * 9.0
    ang = torch.rand(N, generator=g) * 6.28
    asset.data.root_pos_w = ti.env_origins + torch.stack([walked * torch.cos(ang), walked * torch.sin(ang), torch.zeros(N)], dim=-1)
    rec["curr/terrain_origins"], rec["curr/levels_in"], rec["curr/types"] = (ti.terrain_origins.numpy().copy(),
                                                                              ti.terrain_levels.numpy().copy(), ti.terrain_types.numpy().copy())
    rec["curr/env_origins_in"], rec["curr/root_pos_w"], rec["curr/command"] = (ti.env_origins.numpy().copy(),
                                                                                asset.data.root_pos_w.numpy().copy(), command.numpy().copy())
    torch.manual_seed(106)
    mean_level = terrain_levels_vel(env, ids, cfg)
    torch.manual_seed(106)
    rec["curr/randint"] = scatter(N, ids, torch.randint_like(ti.terrain_levels[ids], R)).numpy()
    rec["curr/levels_out"], rec["curr/env_origins_out"] = ti.terrain_levels.numpy().copy(), ti.env_origins.numpy().copy()
    rec["curr/mean_level"] = np.array(float(mean_level), dtype=np.float32)
    rec["meta"] = np.array(json.dumps(dict(N=N, J=J, R=R, C=C, pose_range=pose_range, velocity_range=vel_range, push_range=push_range,
                                           terrain_size=8.0, max_episode_length_s=20.0)))
    np.savez_compressed(os.path.join(GOLDEN, "events.npz"), **rec)
    print("[golden] events:", len(rec), "arrays; reset envs:", k)


if __name__ == "__main__":
    main()
