#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the RSL-RL PPO loop on Isaac-Velocity-Rough-Anymal-C-v0, 4096 envs per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N>1 without a launcher: this process starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...``
     as a CHILD -- it never touches the GPU itself --, relays rank 0's JSON line and exits with the child's code; under torchrun
     (WORLD_SIZE set) it is a rank.  Same contract as the reference's ``python -m torch.distributed.run --nnodes=1 --nproc_per_node=N
     scripts/reinforcement_learning/rsl_rl/train.py --distributed``, docs/source/features/multi_gpu.rst:24-29, train.py:118-126.)

One "step" = one PPO iteration on every rank: 24 x (actor-critic act -> imx_action_process -> [state feed advances]
-> imx_terminations_rewards -> imx_observations(+ray-cast) -> storage) + imx_gae + 5 epochs x 4 minibatches of
(MLP fwd -> imx_ppo_loss -> MLP bwd -> [RCCL all-reduce of the flat gradient bucket] -> imx_adam_step).
value = num_envs x 24 x world_size x K / wall time (max over ranks), inputs resident in HBM before the timed region.

Besides the contract line it reports
  * env_step_path: the post-physics path alone (3 env kernels per step + GAE per 24 steps, no policy) -- the
    like-for-like counterpart of cpu_baseline;
  * roofline: the dominant hand-written kernel (k_obs = observation assembly + fused height-scanner ray-cast),
    algorithmic bytes per launch / average launch duration from HIP events on the launch stream, vs the 8 TB/s peak;
  * cpu_baseline: the CPU restatement of the reference managers + GAE (oracle/, kind "port") on this box's host
    cores, bounded sample, rank 0, N=1 only.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

# Before HIP starts: the PPO update keeps three streams busy (actor, critic, loss values / next gather) and a distributed run adds
# RCCL's.  ROCm's default of 4 hardware queues per process then puts two of the update's streams on one queue as soon as the process
# group exists before them (the order of every multi-GPU launch): 20.2 ms instead of 17.7 ms per update, measured with
# tools/dist_overhead.py.  Eight queues give every stream its own.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TASK = "Isaac-Velocity-Rough-Anymal-C-v0"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # same guide: v_mfma_f32_32x32x2_f32, exact f32 in / f32 accumulate (155 TF measured)


def obs_kernel_bytes_per_env(plan) -> float:
    """ALGORITHMIC HBM bytes of k_obs per env-step, SURVEY.md section 8(d) figures:
    obs-side manager reads (root 13 f, joint pos/vel + defaults 4J f, last action A f, command 3 f) + the obs row
    written (D f) + for the fused z-only ray-cast the unique terrain data under the 1.6 x 1.0 m footprint
    (18 x 12 vertices x 12 B = 2.6 KB; the 187 x 4 B hit-z write of the un-fused form is the obs row itself)."""
    J, A, D, R = plan.num_joints, plan.action_dim, plan.obs_dim, plan.num_rays
    reads = (13 + 4 * J + A + 3) * 4
    mesh = 18 * 12 * 12 if R else 0
    writes = D * 4
    return float(reads + mesh + writes)


def anydrive_like_net(device):
    """A TorchScript-free stand-in for the ANYdrive 3 network of ANYMAL_C_CFG (isaaclab_assets/robots/anymal.py:45-51 downloads the
    real file from Nucleus: absent offline): LSTM(2 -> 8, 2 layers) + 8 -> 16 -> 1 softsign head, seeded random weights."""
    g = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.rand(*s, generator=g) - 0.5).to(device)  # noqa: E731
    lstm = [(r(32, 2), r(32, 8), r(32), r(32)), (r(32, 8), r(32, 8), r(32), r(32))]
    head = [(r(16, 8), r(16)), (r(1, 16), r(1))]
    return lstm, head


def build_env(task, num_envs, device, seed, snapshots, terrain_tiles, mesh=None, full_step=False):
    from isaaclab_amd.env import ManagerBasedRLEnv, load_task_cfg
    from isaaclab_amd.robots import ROBOTS
    from isaaclab_amd.state_feed import StateFeed
    from isaaclab_amd.terrain import make_rough_terrain

    fx = load_task_cfg(task)
    robot = ROBOTS[fx["robot"]]
    terrain = ext = None
    ntri = 0
    if fx["env"]["scene"].get("height_scanner") is not None:
        if mesh is not None:  # reuse an uploaded TerrainMesh (same tiles -> same extent)
            terrain, ext, ntri = mesh, (0.5 * terrain_tiles[0] * 8.0 - 1.0, 0.5 * terrain_tiles[1] * 8.0 - 1.0), mesh.num_triangles
        else:
            v, t, e = make_rough_terrain(terrain_tiles[0], terrain_tiles[1], tile=8.0, horizontal_scale=0.1, border=20.0, seed=0)
            terrain, ext, ntri = (v, t), (e[0] - 1.0, e[1] - 1.0), len(t)
    feed = StateFeed(robot, num_envs, device, seed=seed, num_snapshots=snapshots, extent_xy=ext)
    if not full_step:
        env = ManagerBasedRLEnv(fx, state_feed=feed, terrain=terrain, terrain_cell=0.1, noise_seed=seed)
        return fx, env, ntri
    # --full-step: everything the reference runs in torch around the physics step is the env's own -- command term, contact sensor,
    # reset / interval events, terrain curriculum (imx_reset_orchestrate), ArticulationData refresh, the learned actuator x decimation
    from isaaclab_amd.producers import ActuatorNetLSTM

    env = ManagerBasedRLEnv(fx, state_feed=feed, terrain=terrain, terrain_cell=0.1, noise_seed=seed, own_managers=True, use_contact_sensor=True,
                            use_articulation_update=True)
    if robot.name == "anymal_c":  # ANYDRIVE_3_LSTM_ACTUATOR_CFG (isaaclab_assets/robots/anymal.py:45-51); G1 / Cartpole use implicit actuators
        lstm, head = anydrive_like_net(device)
        env.attach_actuator(ActuatorNetLSTM(num_envs, robot.num_joints, 80.0, 7.5, 120.0, lstm_layers=lstm, head=head, head_activation="softsign",
                                            device=device))
    return fx, env, ntri


def time_env_path(env, runner_storage_T, iters):
    """Post-physics path only: per step 3 env kernels, per T steps one GAE (random rewards in the storage)."""
    from isaaclab_amd.rsl_rl.storage import gae_returns

    N, T = env.num_envs, runner_storage_T
    dev = env.device
    act = torch.randn(N, env.plan.action_dim, device=dev).clamp_(-3, 3)
    rew = torch.randn(T, N, 1, device=dev)
    val = torch.randn(T, N, 1, device=dev)
    dones = torch.zeros(T, N, 1, dtype=torch.uint8, device=dev)
    last = torch.randn(N, 1, device=dev)
    ret, adv = torch.empty_like(rew), torch.empty_like(rew)

    def one_rollout():
        for _ in range(T):
            env.step(act)
        gae_returns(rew, val, dones, last, 0.99, 0.95, True, ret, adv)

    one_rollout()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        one_rollout()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return N * T * iters / dt, dt / (iters * T)


def measured_copy_gbs(device, mib: int = 1024, reps: int = 10) -> float:
    """Device-to-device stream copy (confirm the vendor HBM figure on the box): read + write bytes per second of a
    float4 copy of ``mib`` MiB, HIP events on the current stream."""
    n = mib * 1024 * 1024 // 4
    src = torch.empty(n, device=device).normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize(device)
    return 2.0 * 4.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e9


_blocker = {}


def _events_us(fn, launches, device):
    """Average device time per call of ``fn``: back-to-back launches on torch's current stream, HIP events on that stream.  The host
    needs 5-10 us per call (ctypes, snapshot bookkeeping) -- more than the small kernels run -- so the stream is first kept busy by a
    few milliseconds of device-to-device copies: the launches queue up behind them and then run back to back; the events sit inside
    the queue, after the copies."""
    if device not in _blocker:
        _blocker[device] = (torch.empty(128 * 1024 * 1024, device=device), torch.empty(128 * 1024 * 1024, device=device))
    src, dst = _blocker[device]
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(8):
        fn()
    host_us = (time.perf_counter() - t0) / 8 * 1e6  # what the host needs per call (the launches are asynchronous)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(device)
    for _ in range(max(2, int(launches * host_us * 1.5 / 200.0) + 1)):  # >= 0.2 ms of copy each: the queue never runs dry
        dst.copy_(src)
    e0.record(st)
    for _ in range(launches):
        fn()
    e1.record(st)
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e1) * 1e3 / launches


def time_obs_kernel(env, launches=200):
    """Average duration of the observation kernel ALONE, launched the way the rollout launches it: every launch on the NEXT state
    snapshot of the feed (the 4096 footprints of a snapshot touch ~28 MB of terrain cells: on one unchanged snapshot they would all
    be cache-hot and the figure 15 % too good -- round-2 verdict).  env.step() has the step kernel leave the frame table of the
    state it ran on; to launch the observation kernel by itself each snapshot gets its own scratch block with the frame table
    precomputed (k_frame), and the launches cycle (state pointers, scratch pointer) together."""
    import ctypes

    dev, feed = env.device, env.feed
    S = feed.num_snapshots
    own = env._scratch
    blocks = {}
    for _ in range(S):  # one frame table per snapshot
        feed.advance()
        blocks[feed.index] = own if not blocks else torch.zeros_like(own)
        env._bufs.scratch = blocks[feed.index].data_ptr()
        env._compute_observations()  # k_frame + the observation kernel
    for _ in range(2 * S):
        feed.advance()
        env._bufs.scratch = blocks[feed.index].data_ptr()
        env._compute_observations(frame_current=True)

    def one():
        feed.advance()
        env._bufs.scratch = blocks[feed.index].data_ptr()
        env._compute_observations(frame_current=True, finish_step_tail=env.defer_step_tail)  # the launch env.step() makes (step tail included)

    try:
        us = _events_us(one, launches, dev)
    finally:
        env._bufs.scratch = own.data_ptr()
    return us * 1e-6  # seconds per launch


def term_rew_bytes_per_env(plan, task) -> float:
    """ALGORITHMIC bytes of k_term_rew per env-step.  Headline task: SURVEY.md section 8(d)'s hand count: reads root
    10 f + joint acc / torque 24 f + action / prev 24 f + command 3 f + contact history of the used bodies 81 f + air / contact time 8 f
    + episode length 8 B + episodic sums 11 f; writes reward 1 f + sums 11 f + step_reward 11 f + 3 masks + episode length 8 B + 2
    term_dones = 757 B.  Other tasks: the same count made from the plan's terms."""
    if task == TASK:
        return 757.0
    import numpy as np

    from isaaclab_amd import plan as pm

    w = np.asarray(plan.blob)
    A, H = plan.action_dim, plan.history
    K, NT = int(w[pm.H["NREW"]]), int(w[pm.H["NTERM"]])
    # floats read per id of the term's id list (state arrays touched), by op
    T_, W_ = pm.T_OPS, pm.W_OPS
    per_id_t = {T_["ILLEGAL_CONTACT"]: 3 * H, T_["JOINT_POS_MANUAL_LIMIT"]: 1, T_["JOINT_VEL_LIMIT"]: 2, T_["JOINT_VEL_MANUAL_LIMIT"]: 1,
                T_["JOINT_EFFORT_LIMIT"]: 2}
    per_id_w = {W_["JOINT_TORQUES_L2"]: 1, W_["JOINT_VEL_L1"]: 1, W_["JOINT_VEL_L2"]: 1, W_["JOINT_ACC_L2"]: 1, W_["JOINT_DEVIATION_L1"]: 2,
                W_["JOINT_POS_LIMITS"]: 3, W_["JOINT_VEL_LIMITS"]: 2, W_["APPLIED_TORQUE_LIMITS"]: 2, W_["UNDESIRED_CONTACTS"]: 3 * H,
                W_["CONTACT_FORCES"]: 3 * H, W_["FEET_AIR_TIME"]: 2, W_["FEET_AIR_TIME_POSITIVE_BIPED"]: 2, W_["FEET_SLIDE"]: 3 * H + 3,
                W_["JOINT_POS_TARGET_L2"]: 1, W_["BODY_LIN_ACC_L2"]: 3}
    f = 10 + int(w[pm.H["CMD_DIM"]])  # root quat / lin / ang velocity, command
    for k in range(NT):
        r = w[int(w[pm.H["TERM_OFF"]]) + k * pm.REC_WORDS:][:pm.REC_WORDS]
        f += per_id_t.get(int(r[pm.R["OP"]]), 0) * int(r[pm.R["NIDS"]])
    for k in range(K):
        r = w[int(w[pm.H["REW_OFF"]]) + k * pm.REC_WORDS:][:pm.REC_WORDS]
        if int(r[pm.R["WEIGHT"]]) == 0:  # +0.0f: skipped at run time
            continue
        op = int(r[pm.R["OP"]])
        f += 2 * A if op == W_["ACTION_RATE_L2"] else (A if op == W_["ACTION_L2"] else per_id_w.get(op, 0) * int(r[pm.R["NIDS"]]))
    reads = 4 * (f + K) + 8    # + episodic sums, episode length
    writes = 4 * (1 + 2 * K) + 3 + 8 + NT  # reward, sums, step_reward, three masks, episode length, term_dones
    return float(reads + writes)


def time_step_kernels(env, task, T, launches=200):
    """roofline_step: every hand-written kernel of the post-physics path as the rollout runs it -- back-to-back launches of ONE kernel,
    each on the next state snapshot, HIP events on the launch stream -- with its algorithmic bytes against the
    8 TB/s peak.  k_term_rew is launched with the rollout slot (imx_terminations_rewards_rollout), as the three-launch rollout step does."""
    import ctypes

    from isaaclab_amd import _lib
    from isaaclab_amd._lib import ImxRolloutSlot, check
    from isaaclab_amd.rsl_rl.storage import gae_returns

    dev, N, plan = env.device, env.num_envs, env.plan
    L, st = env._lib, _lib.current_stream(env.device)
    A = plan.action_dim
    act = torch.randn(N, max(A, 1), device=dev).clamp_(-3, 3)
    val, rew_o = torch.randn(N, 1, device=dev), torch.empty(N, 1, device=dev)
    dones_o = torch.empty(N, 1, dtype=torch.uint8, device=dev)
    cur_r, cur_l, eps = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.zeros(3, device=dev)
    slot = ImxRolloutSlot(value_t=val.data_ptr(), rewards_out=rew_o.data_ptr(), dones_out=dones_o.data_ptr(), cur_reward_sum=cur_r.data_ptr(),
                          cur_ep_len=cur_l.data_ptr(), ep_stats3=eps.data_ptr(), gamma=0.99, bootstrap_time_outs=1)

    def k_action():
        env.feed.advance()
        env._process_action(act)

    def k_term_rew():
        env.feed.advance()
        check(L.imx_terminations_rewards_rollout(env._plan_h, N, ctypes.byref(env._state()), ctypes.byref(env._bufs), 1, ctypes.byref(slot), st))

    out = {}

    def entry(name, us, nbytes, **kw):
        gbs = nbytes / (us * 1e-6) / 1e9
        out[name] = {"bytes_per_launch": nbytes, "avg_launch_us": us, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, **kw}

    for _ in range(3):
        env.step(act)
    if A > 0:
        for _ in range(8):
            k_action()
        entry("k_action", _events_us(k_action, launches, dev), 6.0 * A * 4 * N,
              what="imx_action_process alone (env.step()); in the rollout its work is the epilogue of k_mlp_infer")
    for _ in range(8):
        k_term_rew()
    entry("k_term_rew", _events_us(k_term_rew, launches, dev), term_rew_bytes_per_env(plan, task) * N,
          what="terminations + rewards + reset bookkeeping + storage slot t (rewards, dones, episode statistics); the 128 B frame / sensor "
               "row it leaves per env is not in the algorithmic figure")
    k_s = time_obs_kernel(env, launches)
    entry(L.imx_observations_kernel_name(env._plan_h).decode(), k_s * 1e6, obs_kernel_bytes_per_env(plan) * N,
          what="observation assembly" + (" + fused height-scanner ray-cast" if plan.num_rays else ""))
    # GAE: k_gae + k_adv_normalize on storage-shaped tensors (25 B per transition: r 4, V 4, done 1, ret 4, adv 4 + normalise 8)
    rew, values = torch.randn(T, N, 1, device=dev), torch.randn(T, N, 1, device=dev)
    dn = (torch.rand(T, N, 1, device=dev) < 0.02).to(torch.uint8)
    last = torch.randn(N, 1, device=dev)
    ret, adv = torch.empty_like(rew), torch.empty_like(rew)
    scr = torch.zeros(int(L.imx_gae_scratch_bytes(T, N)), dtype=torch.uint8, device=dev)

    def gae():
        gae_returns(rew, values, dn, last, 0.99, 0.95, True, ret, adv, scr)

    for _ in range(8):
        gae()
    entry("k_gae + k_adv_normalize", _events_us(gae, launches, dev), 25.0 * T * N, what=f"one rollout of T = {T} steps (two launches)")
    return out


def time_producers(env, launches=200):
    """roofline_producers: the SURVEY 8(f) kernels of the full step, each as back-to-back launches on the next state snapshot, with its
    algorithmic bytes per env against the 8 TB/s peak."""
    dev, N, f = env.device, env.num_envs, env.feed
    J, B = env.plan.num_joints, env.plan.robot.num_bodies
    out = {}

    def entry(name, us, per_env, **kw):
        gbs = per_env * N / (us * 1e-6) / 1e9
        out[name] = {"bytes_per_launch": per_env * N, "avg_launch_us": us, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, **kw}

    cs, ar, an = env.contact_sensor, env.articulation, env.actuator_net
    if cs is not None:
        H = cs.history_length

        def contact():
            f.advance()
            cs.update(f["net_forces_w_history"][:, 0], env.step_dt)

        for _ in range(8):
            contact()
        entry("k_contact_update + k_contact_stamp", _events_us(contact, launches, dev), 4.0 * (3 * B + (2 * H - 1) * 3 * B + 8 * B + 4) + 2,
              what="ContactSensor.update: new forces, history shift, air / contact times (two launches)")
    if env._has_orchestration:
        def orch():
            f.advance()
            env._orchestrate(env.reset_buf, do_step=True)

        for _ in range(8):
            orch()
        entry("k_reset_orchestrate", _events_us(orch, launches, dev), 4.0 * (2 * 9 + 10 + 2 + 3) + 2 * 8 + 4,
              what="_reset_idx of the reset envs + CommandManager.compute + interval events; bytes: command state read + written (9 f), root "
                   "state 10 f, timers, origins, level, counter, masks (event bodies touch the firing envs only)")
    if ar is not None:
        def artic():
            f.advance()
            ar.update(f.physx("root_transforms"), f.physx("root_velocities"), f["joint_vel"], env.step_dt)

        for _ in range(8):
            artic()
        entry("k_articulation_update", _events_us(artic, launches, dev), 4.0 * (13 + J + 13 + 2 * J),
              what="ArticulationData: root state re-laid (XYZW -> WXYZ), joint_acc finite difference")
    if an is not None:
        tgt = torch.randn(N, J, device=dev)

        def lstm():
            f.advance()
            an.compute(tgt, f["joint_pos"], f["joint_vel"])

        for _ in range(8):
            lstm()
        L, Hd = an.num_layers, an.hidden_dim
        entry("k_actuator_net_lstm", _events_us(lstm, launches, dev), 4.0 * J * (3 + 4 * L * Hd + 2),
              what=f"ActuatorNetLSTM.compute, one launch per physics substep ({int(env.cfg_decimation)} per env step): per (env, joint) sample the "
                   f"LSTM({L} x {Hd}) state read + written, targets, efforts",
              flops_per_launch=2.0 * N * J * (4 * Hd * (2 + Hd) + (L - 1) * 4 * Hd * 2 * Hd + Hd * 16 + 16))
    return out


def cpu_baseline(task, num_envs, T, budget_s=12.0):
    """oracle/ restatement of TerminationManager + RewardManager + ObservationManager (+ action affine) + GAE on
    torch-CPU, same synthetic feed distribution (height-scan hits supplied; no ray-cast on the CPU side)."""
    from isaaclab_amd.env import load_task_cfg
    from isaaclab_amd.robots import ROBOTS
    from isaaclab_amd.state_feed import StateFeed
    from oracle.mdp_oracle import OracleEnv
    from oracle.rsl_rl_oracle import compute_returns

    fx = load_task_cfg(task)
    robot = ROBOTS[fx["robot"]]
    feed = StateFeed(robot, num_envs, "cpu", seed=42, num_snapshots=4)
    env = OracleEnv(fx["env"], robot.joint_names, robot.body_names, num_envs, feed.__getitem__, feed.gravity_dir)
    g = torch.Generator().manual_seed(0)
    if fx["env"]["scene"].get("height_scanner") is not None:
        env.ray_hits_w = torch.rand(num_envs, 187, 3, generator=g) * 0.2
    act = torch.randn(num_envs, env.A, generator=g).clamp_(-3, 3)
    rew, val = torch.randn(T, num_envs, 1, generator=g), torch.randn(T, num_envs, 1, generator=g)
    dones = torch.zeros(T, num_envs, 1, dtype=torch.uint8)
    last = torch.randn(num_envs, 1, generator=g)
    env.episode_length_buf[:] = torch.randint(0, env.max_episode_length, (num_envs,), generator=g)

    def step():
        env.process_action(act)
        feed.advance()
        env.post_physics_step(None)

    def rollout():
        for _ in range(T):
            step()
        compute_returns(rew, val, dones, last, 0.99, 0.95, True)

    # eager torch on small tensors is dispatch-bound: more threads is not faster.  Probe a few thread counts on a
    # short sample, then spend the budget on the best one (the GPU box gives 16 cores to a 1-GPU job).
    ncpu = os.cpu_count() or 1
    cands = sorted({1, 4, 8, min(16, ncpu)})
    probe = {}
    for nt in cands:
        torch.set_num_threads(nt)
        rollout()
        t0 = time.perf_counter()
        rollout()
        probe[nt] = time.perf_counter() - t0
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        rollout()
        steps += T
    dt = time.perf_counter() - t0
    probe_s = ", ".join(f"{k} thr: {num_envs * T / v:.0f}/s" for k, v in probe.items())
    return {"value": num_envs * steps / dt, "unit": "env-steps/s", "cores": best, "kind": "port",
            "sample": f"{steps} env-steps of {num_envs} envs ({steps // T} rollouts incl. GAE, obs noise on, "
                      f"height-scan hits supplied, no policy), {dt:.1f} s, torch {torch.__version__} CPU, "
                      f"{ncpu} host cores visible; probe {probe_s}"}


def time_mlp_dw(alg, M: int, device, launches: int = 30) -> dict:
    """imx_mlp_dw (k_mlp_dw + k_mlp_reduce) on the hidden-layer shapes of the actor at the minibatch size, isolated
    launches on the current stream, HIP events around each shape's batch.  flops = 2*M*out*in per launch."""
    from isaaclab_amd._lib import check, current_stream, lib
    from isaaclab_amd.rsl_rl.ppo import HEAD_MAX_OUT

    L = lib()
    st = current_stream(device)
    flops = 0.0
    secs = 0.0
    secs_k = 0.0
    shapes = []
    for lin, _ in alg._actor_layers:
        N, K = lin.out_features, lin.in_features
        if N <= HEAD_MAX_OUT:
            continue
        dY = torch.randn(M, N, device=device)
        ldx = (K + 3) // 4 * 4  # rows on 16-byte boundaries, as the minibatch buffers of the update keep them (storage.py)
        X = torch.randn(M, ldx, device=device)
        dW, db = torch.empty(N, K, device=device), torch.empty(N, device=device)
        nb = int(L.imx_mlp_scratch_bytes(M, N, K))
        scr = torch.empty(nb, dtype=torch.uint8, device=device)
        args = (M, N, K, dY.data_ptr(), N, X.data_ptr(), ldx, dW.data_ptr(), db.data_ptr(), scr.data_ptr(), nb, st)
        for _ in range(3):
            check(L.imx_mlp_dw(*args))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            check(L.imx_mlp_dw(*args))
        b.record()
        torch.cuda.synchronize(device)
        t = a.elapsed_time(b) * 1e-3 / launches
        # ... and k_mlp_dw ALONE, as the update runs it: the reductions of a backward pass are deferred to one batched launch
        # (imx_reduce_batch_*), so the events bracket eight launches whose reductions are only queued; the flush follows the end event
        import ctypes

        h = ctypes.c_void_p()
        check(L.imx_reduce_batch_create(ctypes.byref(h)))
        t_k = 0.0
        for _ in range(4):
            check(L.imx_reduce_batch_begin(h))
            a.record()
            for _ in range(8):
                check(L.imx_mlp_dw(*args))
            b.record()
            check(L.imx_reduce_batch_flush(h, st))
            torch.cuda.synchronize(device)
            t_k += a.elapsed_time(b) * 1e-3 / 32
        L.imx_reduce_batch_destroy(h)
        shapes.append({"out": N, "in": K, "us": t * 1e6, "tflops": 2.0 * M * N * K / t / 1e12, "kernel_only_us": t_k * 1e6,
                       "kernel_only_tflops": 2.0 * M * N * K / t_k / 1e12})
        flops += 2.0 * M * N * K
        secs += t
        secs_k += t_k
    n = max(len(shapes), 1)
    return {"bound": "mfma", "kernel": "k_mlp_dw + k_mlp_reduce (imx_mlp_dw: dW = dY^T X, db; hidden layers of the actor, isolated launches)",
            "achieved": flops / secs / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / secs / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "flops_per_launch": flops / n, "avg_launch_us": secs / n * 1e6, "samples": M, "per_layer": shapes,
            "kernel_only": {"what": "k_mlp_dw without its reduction launch (deferred and batched in the update)", "achieved": flops / secs_k / 1e12,
                            "frac": flops / secs_k / 1e12 / MFMA_F32_PEAK_TFLOPS, "avg_launch_us": secs_k / n * 1e6}}


def self_launch(n_gpus: int) -> int:
    """``python bench.py --gpus N`` (N > 1) outside a launcher: run the N ranks as a child ``torch.distributed.run`` and relay rank 0's
    line.  This parent has not initialised HIP (importing torch does not) and never does: the children are fresh processes."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:  # rank 0 prints exactly one JSON line on fd 1; anything else a library wrote there goes to stderr
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but rank 0 printed no result line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--task", default=TASK)
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--snapshots", type=int, default=4)
    ap.add_argument("--terrain-tiles", type=int, nargs=2, default=(10, 20))
    ap.add_argument("--full-step", action="store_true", help="the env owns everything the reference runs in torch around the physics step "
                    "(command term, contact sensor, reset / interval events, terrain curriculum, ArticulationData refresh, ActuatorNetLSTM x "
                    "decimation): all of it inside the captured rollout; reported as env_step_path_full + roofline_producers")
    ap.add_argument("--no-graph", action="store_true", help="eager rollout instead of one hipGraph replay per rollout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-n", action="store_true", help="skip the 65 536-env launches of the observation kernel (they share "
                    "its name and would skew a rocprofv3 --stats average taken over this command)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # before anything touches the GPU
        raise SystemExit(self_launch(args.gpus))

    # stdout carries exactly ONE line (the JSON): native libraries on the GPU box print to fd 1 while the device is
    # initialised (libdrm's "amdgpu.ids: No such file or directory"), so fd 1 points at stderr until the final print
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (python bench.py --gpus N does it itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libimx has no CPU path")
    # IMX_REHEARSE_ONE_GPU=1 (with torchrun --nproc-per-node 2): every rank on cuda:0 and a gloo group -- the whole multi-rank
    # flow of this file (barriers, max-over-ranks timing, rank-0 line, tear-down) and of PPO.update on a one-GPU box; RCCL itself is
    # exercised by IMX_FORCE_DIST=1 (single-rank nccl group) and tests/test_kernels_gpu.py
    rehearsal = os.environ.get("IMX_REHEARSE_ONE_GPU") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist

    force_dist = os.environ.get("IMX_FORCE_DIST") == "1"  # exercise the RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

    # ---- self-diagnosis of a multi-rank launch (the first N > 1 hardware run is the driver's): fail loudly on a broken rendezvous
    def pci_id(i):
        p = torch.cuda.get_device_properties(i)
        if hasattr(p, "pci_bus_id"):
            return "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
        return str(getattr(p, "uuid", i))

    my_gpu = pci_id(dev_index)
    if world > 1 or force_dist:
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: the process group has {dist.get_world_size()} ranks but --gpus is {args.gpus}: start exactly one rank per GPU "
                             "(python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N, or plain python bench.py --gpus N)")
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, my_gpu)
        if not rehearsal and len(set(ids)) != len(ids):
            raise SystemExit(f"bench.py: two ranks report the same GPU (PCI ids {ids}): LOCAL_RANK / device visibility is wrong -- every rank would "
                             "time the same card.  (IMX_REHEARSE_ONE_GPU=1 allows it on purpose: a one-GPU rehearsal over gloo.)")
    else:
        ids = [my_gpu]

    torch.manual_seed(42 + rank)  # (scripts/reinforcement_learning/rsl_rl/train.py:118-126: seed + local rank, for diversity across ranks)
    t_build = time.perf_counter()
    fx, env, ntri = build_env(args.task, args.num_envs, device, 42 + rank, args.snapshots, tuple(args.terrain_tiles), full_step=args.full_step)
    torch.cuda.synchronize(device)
    terrain_build_s = time.perf_counter() - t_build  # terrain generation + mesh / grid build + state feed + env buffers, per rank
    agent = fx["agent"]
    venv = RslRlVecEnvWrapper(env, clip_actions=agent.get("clip_actions"))
    runner = OnPolicyRunner(venv, agent, log_dir=None, device=str(device), use_graph=not args.no_graph)
    T = runner.num_steps_per_env
    venv.episode_length_buf = torch.randint_like(venv.episode_length_buf, high=int(venv.max_episode_length))
    if world > 1 or force_dist:
        runner.alg.broadcast_parameters()
    runner.train_mode()

    def iteration():
        runner.collect()
        with torch.inference_mode():
            runner.alg.compute_returns(runner.last_obs)
        runner.alg.update()

    # untimed priming, whatever --warmup says: the rollout graph is captured in the first iteration and the update decides between
    # eager issue and hipGraph replay over its first five calls (PPO.update) -- neither may fall into the timed region
    t_cap = time.perf_counter()
    iteration()  # the first one warms up and captures the rollout graph
    torch.cuda.synchronize(device)
    graph_capture_s = time.perf_counter() - t_cap
    for _ in range(4):
        iteration()
    for _ in range(args.warmup):
        iteration()
    runner.alg.time_allreduce = world > 1 or force_dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):  # same work as iteration(), with three device-side event records (no host syncs)
        ev[3 * k].record()
        runner.collect()
        with torch.inference_mode():
            runner.alg.compute_returns(runner.last_obs)
        ev[3 * k + 1].record()
        runner.alg.update()
        ev[3 * k + 2].record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    rank_ms = [1e3 * elapsed / args.steps]
    ar = runner.alg.allreduce_us()
    per_rank = {"terrain_build_s": [terrain_build_s], "graph_capture_s": [graph_capture_s], "allreduce_us_mean": [ar["mean"] if ar else None]}
    if world > 1:
        rows = [None] * world
        dist.all_gather_object(rows, (terrain_build_s, graph_capture_s, ar["mean"] if ar else None))
        per_rank = {"terrain_build_s": [r[0] for r in rows], "graph_capture_s": [r[1] for r in rows], "allreduce_us_mean": [r[2] for r in rows]}
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        rank_ms = [1e3 * float(x.item()) / args.steps for x in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = args.num_envs * T * world * args.steps / elapsed
    stats = runner.alg.loss_dict()
    collect_ms = sum(ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(args.steps)) / args.steps
    update_ms = sum(ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(args.steps)) / args.steps

    # --- secondary measurements on rank 0 (outside the timed region)
    out = {
        "metric": ("env-steps/sec (whole node), Anymal-C rough 4096 envs/GPU, RSL-RL PPO iteration (collect+GAE+update)"
                   if args.task == TASK and args.num_envs == 4096 else
                   f"env-steps/sec (whole node), {args.task} {args.num_envs} envs/GPU, RSL-RL PPO iteration"),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.task}, {args.num_envs} envs/GPU, T={T}, 5 epochs x 4 minibatches, "
                               f"terrain {ntri} triangles, {args.snapshots} state snapshots resident in HBM",
                   "parallelism": f"dp{world}", "rollout": "eager" if args.no_graph else "hipGraph",
                   "full_step": bool(args.full_step),
                   "update": ("hipGraph" if getattr(runner.alg, "_update_g", None) is not None else "eager")
                   + (" (measured eager %.2f ms, graph %.2f ms)" % runner.alg._update_times_ms if hasattr(runner.alg, "_update_times_ms") else ""),
                   "policy_params": runner.alg.bucket.numel,
                   "obs_noise": "in-kernel counter-based generator (range / determinism property-tested; the parity tests feed the "
                                "reference's recorded uniforms through the same kernel)"},
        "phase_ms": {"collect_plus_gae": collect_ms, "update": update_ms},
        "ms_per_step_per_rank": rank_ms,
        "per_rank": per_rank,
        "collective": ({"backend": dist.get_backend(), "ranks": dist.get_world_size(), "distinct_gpus": len(set(ids)), "gpu_pci_ids": ids,
                        "allreduce_us": ar,  # rank 0: HIP events around each bucket all-reduce on the update's stream (incl. waiting for peers)
                        "grad_allreduce_per_iteration": int(runner.alg.num_learning_epochs) * int(runner.alg.num_mini_batches),
                        "bucket_bytes": 4 * (runner.alg.bucket.numel + 8)} if (world > 1 or force_dist) else None),
    }
    if rank == 0:
        env_rate, env_step_s = time_env_path(env, T, iters=20)
        out["env_step_path_full" if args.full_step else "env_step_path"] = {
            "value": env_rate, "unit": "env-steps/s", "us_per_env_step_batch": env_step_s * 1e6,
            "what": ("imx_action_process + ActuatorNetLSTM x decimation + imx_articulation_update + imx_contact_sensor_update + "
                     "imx_terminations_rewards + imx_reset_orchestrate + imx_observations per step, imx_gae per 24 steps; no policy")
            if args.full_step else "imx_action_process + imx_terminations_rewards + imx_observations per step, imx_gae per 24 steps; no policy"}
        if args.full_step:
            out["roofline_producers"] = time_producers(env)
        steps = time_step_kernels(env, args.task, T)
        obs_name = env._lib.imx_observations_kernel_name(env._plan_h).decode()
        ko = steps[obs_name]
        traffic = traffic_src = None
        tf = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")  # PMC passes of tools/pmc_obs.py (same config), see profiles/README.md
        if os.path.exists(tf) and args.num_envs == 4096 and args.task == TASK:
            traffic = json.load(open(tf)).get("k_obs_bytes_per_launch")
            traffic_src = ("profiles/r03_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_obs.py on the same "
                           "configuration, committed; NOT observed by this run)")
        out["roofline"] = {"bound": "hbm", "kernel": obs_name + " (" + ko["what"] + ")",
                           "achieved": ko["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ko["frac"],
                           "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": ko["bytes_per_launch"],
                           "avg_launch_us": ko["avg_launch_us"],
                           "how": f"HIP events around 200 back-to-back launches, each on the next of the feed's {args.snapshots} state snapshots "
                                  "(as the rollout runs it)",
                           "peak_measured_copy": measured_copy_gbs(device)}  # GB/s of a 1 GiB device-to-device copy on this box
        out["roofline_step"] = steps
        # the same kernel at 16x the batch (65 536 envs): what the layout reaches once launch latency is amortised
        if os.environ.get("IMX_BENCH_LARGE_N", "1") == "1" and not args.no_large_n and args.task == TASK:
            try:
                big_n = 65536
                _, env_big, _ = build_env(args.task, big_n, device, 7, 1, tuple(args.terrain_tiles), mesh=env.terrain)
                env_big.reset()
                kb = time_obs_kernel(env_big, launches=48)
                out["roofline_large_n"] = {"num_envs": big_n, "kernel": obs_name, "avg_launch_us": kb * 1e6,
                                           "achieved": obs_kernel_bytes_per_env(env_big.plan) * big_n / kb / 1e9,
                                           "frac": obs_kernel_bytes_per_env(env_big.plan) * big_n / kb / 1e9 / HBM_PEAK_GBS,
                                           "unit": "GB/s"}
                del env_big
            except Exception as exc:  # secondary measurement only
                out["roofline_large_n"] = {"error": str(exc)}
        try:  # the largest hand-written kernel of the update, against the f32 MFMA peak
            mb = args.num_envs * T // int(runner.alg.num_mini_batches)
            out["roofline_mfma"] = time_mlp_dw(runner.alg, mb, device)
            tf1 = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")  # k_mlp_dw is unchanged since round 1
            if os.path.exists(tf1):
                out["roofline_mfma"]["traffic"] = json.load(open(tf1)).get("k_mlp_dw_bytes_per_launch")
        except Exception as exc:  # secondary measurement only
            out["roofline_mfma"] = {"error": str(exc)}
        out["ppo"] = {k: round(v, 6) for k, v in stats.items()}
        out["ppo"]["learning_rate"] = runner.alg.learning_rate
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.task, args.num_envs, T, args.cpu_budget)
            if "env_step_path" in out:
                out["env_step_path"]["vs_cpu_baseline"] = env_rate / out["cpu_baseline"]["value"]
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
