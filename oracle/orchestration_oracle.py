"""TEST INFRASTRUCTURE -- CPU restatement of the reference's reset / interval ORCHESTRATION around the env step,
pinned by tests/golden/orchestration.npz (oracle/gen_golden_orchestration.py: the REAL ``ManagerBasedRLEnv._reset_idx``, EventManager,
CommandManager and CurriculumManager).  Never imported by the product.

* ``_reset_idx``                       isaaclab/isaaclab/envs/manager_based_rl_env.py:347-392
* the tail of ``step``                 :232-236 (command_manager.compute, event_manager.apply("interval"))
* ``EventManager.apply`` / ``reset``   isaaclab/isaaclab/managers/event_manager.py:123-148,150-273
* ``CommandManager.reset``             isaaclab/isaaclab/managers/command_manager.py:340-358 (+ CommandTerm.reset :119-147)
* ``CurriculumManager.reset``          isaaclab/isaaclab/managers/curriculum_manager.py:95-118
* the event terms / curriculum term    oracle/events_oracle.py; the command term: oracle/producers_oracle.py

Uniform samples are inputs: per event term (N, width) rows, per interval term (N) for the timer re-sampling (a global timer takes [0]),
(2, N, 7) for the command term, (N) random terrain levels.  ``asset.data`` is the state feed; what the terms would write to the simulator
goes to ``sim_writes`` (persistent (N, .) buffers, rows of the affected envs).
"""

from __future__ import annotations

import torch

from . import events_oracle as ev
from .producers_oracle import VelocityCommandOracle

AXES = ev.AXES


def _fn(term: dict) -> str:
    f = term.get("func", "")
    f = f if isinstance(f, str) else getattr(f, "__name__", "")
    return f.replace(":", ".").rsplit(".", 1)[-1]


class OrchestrationOracle:
    def __init__(self, env_cfg: dict, num_envs: int, num_joints: int, body_names: list, step_dt: float, max_episode_length_s: float,
                 default_root_state, default_joint_pos, default_joint_vel, soft_joint_pos_limits, soft_joint_vel_limits,
                 terrain_origins, terrain_levels, terrain_types, terrain_size_x: float, interval_time_left_init):
        N = self.N = num_envs
        self.J, self.body_names = num_joints, list(body_names)
        self.step_dt, self.max_len_s = step_dt, max_episode_length_s
        self.drs, self.djp, self.djv, self.plim, self.vlim = default_root_state, default_joint_pos, default_joint_vel, soft_joint_pos_limits, soft_joint_vel_limits
        self.terms = [(name, t) for name, t in (env_cfg.get("events") or {}).items() if t is not None and t.get("mode") in ("reset", "interval")]
        self.reset_terms = [(n, t) for n, t in self.terms if t["mode"] == "reset"]
        self.interval_terms = [(n, t) for n, t in self.terms if t["mode"] == "interval"]
        # EventManager state (event_manager.py:_prepare_terms)
        self.last_triggered = {n: torch.zeros(N, dtype=torch.int32) for n, _ in self.reset_terms}
        self.triggered_once = {n: torch.zeros(N, dtype=torch.bool) for n, _ in self.reset_terms}
        self.time_left = {}
        for i, (n, t) in enumerate(self.interval_terms):
            tl = interval_time_left_init[i].clone()
            self.time_left[n] = tl[:1].clone() if t.get("is_global_time") else tl
        B = len(body_names)
        self.sim_writes = {"root_pose": torch.zeros(N, 7), "root_vel": torch.zeros(N, 6), "joint_pos": torch.zeros(N, num_joints),
                           "joint_vel": torch.zeros(N, num_joints), "ext_force": torch.zeros(N, B, 3), "ext_torque": torch.zeros(N, B, 3)}
        # terrain curriculum (TerrainImporter state)
        self.terrain_origins, self.levels, self.types = terrain_origins, terrain_levels.clone(), terrain_types
        self.env_origins = terrain_origins[self.levels, self.types].clone()
        self.size_x = terrain_size_x
        self.has_curriculum = any(v is not None for v in (env_cfg.get("curriculum") or {}).values())
        self.curriculum_names = [k for k, v in (env_cfg.get("curriculum") or {}).items() if v is not None]
        cname, ccfg = next(iter(env_cfg["commands"].items()))
        self.command_name = cname
        self.cmd = VelocityCommandOracle(ccfg, N, step_dt)
        self.log: dict = {}

    # ---- one event term on the envs `ids`
    def _run_term(self, name, t, ids, feed, U):
        fn, p = _fn(t), t.get("params", {}) or {}
        u = U[name]
        if len(ids) == 0:
            return
        if fn == "reset_root_state_uniform":
            pose, vel = ev.reset_root_state_uniform(self.drs[ids], self.env_origins[ids], p.get("pose_range") or {}, p.get("velocity_range") or {},
                                                    u[ids, 0:6], u[ids, 6:12])
            self.sim_writes["root_pose"][ids], self.sim_writes["root_vel"][ids] = pose, vel
        elif fn in ("reset_joints_by_scale", "reset_joints_by_offset"):
            J = self.J
            pos, vel = ev.reset_joints(self.djp[ids], self.djv[ids], self.plim[ids], self.vlim[ids], tuple(p["position_range"]),
                                       tuple(p["velocity_range"]), u[ids, :J], u[ids, J:2 * J], by_offset=fn.endswith("offset"))
            self.sim_writes["joint_pos"][ids], self.sim_writes["joint_vel"][ids] = pos, vel
        elif fn == "push_by_setting_velocity":
            root_vel_w = torch.cat([feed["root_lin_vel_w"], feed["root_ang_vel_w"]], dim=-1)  # asset.data.root_vel_w
            self.sim_writes["root_vel"][ids] = ev.push_by_setting_velocity(root_vel_w[ids], p.get("velocity_range") or {}, u[ids, :6])
        elif fn == "apply_external_force_torque":
            import re

            names = (p.get("asset_cfg") or {}).get("body_names")
            keys = [names] if isinstance(names, str) else names
            bids = list(range(len(self.body_names))) if names is None else [i for i, b in enumerate(self.body_names) if any(re.fullmatch(k, b) for k in keys)]
            nb = len(bids)
            f, tq = ev.apply_external_force_torque(tuple(p["force_range"]), tuple(p["torque_range"]), u[ids, :3 * nb].reshape(-1, nb, 3),
                                                   u[ids, 3 * nb:6 * nb].reshape(-1, nb, 3))
            for k, b in enumerate(bids):
                self.sim_writes["ext_force"][ids, b], self.sim_writes["ext_torque"][ids, b] = f[:, k], tq[:, k]
        else:
            raise NotImplementedError(fn)

    # ---- ManagerBasedRLEnv._reset_idx (manager part the producers own)
    def reset_idx(self, ids, feed, step_count: int, U, U_cmd, rand_levels):
        mask = torch.zeros(self.N, dtype=torch.bool)
        mask[ids] = True
        log = {}
        if self.has_curriculum:  # curriculum_manager.compute(env_ids): terrain_levels_vel -> update_env_origins
            lv, org, mean = ev.terrain_levels_vel(mask, feed["root_pos_w"], self.env_origins, self.cmd.vel_command_b, self.terrain_origins, self.levels,
                                                  self.types, self.size_x, self.max_len_s, rand_levels)
            self.levels, self.env_origins = lv, org
            for n in self.curriculum_names:
                log[f"Curriculum/{n}"] = float(mean)
        # event_manager.apply("reset", env_ids, global_env_step_count) (event_manager.py:233-260)
        for name, t in self.reset_terms:
            msc = int(t.get("min_step_count_between_reset", 0) or 0)
            if msc == 0:
                self.last_triggered[name][ids] = step_count
                self.triggered_once[name][ids] = True
                valid = ids
            else:
                last, once = self.last_triggered[name][ids], self.triggered_once[name][ids]
                trig = (step_count - last >= msc) | ((last == 0) & ~once)
                valid = ids[trig]
                self.triggered_once[name][valid] = True
                self.last_triggered[name][valid] = step_count
            self._run_term(name, t, valid, feed, U)
        # command_manager.reset(env_ids): metrics logged and zeroed, counters zeroed, resample (command_manager.py:119-147,340-358)
        c = self.cmd
        for m in ("error_vel_xy", "error_vel_yaw"):
            log[f"Metrics/{self.command_name}/{m}"] = float(torch.mean(c.metrics[m][ids]))
            c.metrics[m][ids] = 0.0
        c.command_counter[ids] = 0
        c._resample(ids, U_cmd)
        # event_manager.reset(env_ids): function terms keep their interval timers (:123-148 walks the CLASS terms only)
        self.log = log
        return log

    # ---- the tail of ManagerBasedRLEnv.step (:232-236) after the (optional) _reset_idx
    def step_tail(self, feed, U, U_int, U_cmd):
        c = self.cmd
        # command_manager.compute(dt)  (the draw counters of this step continue: a reset env's timer resample is draw 1)
        ident = torch.zeros(self.N, dtype=torch.bool)
        draw = c._draw.clone()
        c.reset_and_compute(self.step_dt, feed["root_quat_w"], feed["root_lin_vel_w"], feed["root_ang_vel_w"], ident, _KeepDraw(U_cmd, draw, c))
        # event_manager.apply("interval", dt)  (event_manager.py:205-232)
        fired = {}
        for i, (name, t) in enumerate(self.interval_terms):
            lo, hi = t["interval_range_s"]
            tl = self.time_left[name]
            tl -= self.step_dt
            if t.get("is_global_time"):
                if tl < 1e-6:
                    tl[:] = U_int[i][:1] * (hi - lo) + lo
                    ids = torch.arange(self.N)
                    self._run_term(name, t, ids, feed, U)
                    fired[name] = ids
            else:
                ids = (tl < 1e-6).nonzero().flatten()
                if len(ids) > 0:
                    tl[ids] = U_int[i][ids] * (hi - lo) + lo
                    self._run_term(name, t, ids, feed, U)
                    fired[name] = ids
        return fired


class _KeepDraw:
    """``VelocityCommandOracle.reset_and_compute`` zeroes its per-call draw counters first; inside a step the command term's reset (in
    ``_reset_idx``) and its compute are two calls on the SAME table, the timer resample of a reset env being draw 1: this view re-adds
    the draws already taken."""

    def __init__(self, U, taken, cmd):
        self.U, self.taken = U, taken

    def __getitem__(self, key):
        d, ids, col = key
        return self.U[d + self.taken[ids], ids, col]
