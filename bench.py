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


def build_env(task, num_envs, device, seed, snapshots, terrain_tiles, mesh=None):
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
    env = ManagerBasedRLEnv(fx, state_feed=feed, terrain=terrain, terrain_cell=0.1, noise_seed=seed)
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


def time_obs_kernel(env, launches=200):
    """Average duration of k_obs from HIP events on the stream it is launched on (torch's current stream)."""
    dev = env.device
    for _ in range(10):
        env._compute_observations()  # k_frame + the observation kernel: the frame table of this state is current from here on
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(launches):
        env._compute_observations(frame_current=True)  # k_obs alone, as in env.step()
    e1.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e-3 / launches  # seconds per launch


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
        shapes.append({"out": N, "in": K, "us": t * 1e6, "tflops": 2.0 * M * N * K / t / 1e12})
        flops += 2.0 * M * N * K
        secs += t
    n = max(len(shapes), 1)
    return {"bound": "mfma", "kernel": "k_mlp_dw + k_mlp_reduce (imx_mlp_dw: dW = dY^T X, db; hidden layers of the actor, isolated launches)",
            "achieved": flops / secs / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / secs / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "flops_per_launch": flops / n, "avg_launch_us": secs / n * 1e6, "samples": M, "per_layer": shapes}


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

    torch.manual_seed(42 + rank)
    fx, env, ntri = build_env(args.task, args.num_envs, device, 42 + rank, args.snapshots, tuple(args.terrain_tiles))
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
    for _ in range(5):
        iteration()
    for _ in range(args.warmup):
        iteration()
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
                   "update": ("hipGraph" if getattr(runner.alg, "_update_g", None) is not None else "eager")
                   + (" (measured eager %.2f ms, graph %.2f ms)" % runner.alg._update_times_ms if hasattr(runner.alg, "_update_times_ms") else ""),
                   "policy_params": runner.alg.bucket.numel,
                   "obs_noise": "in-kernel counter-based generator (range / determinism property-tested; the parity tests feed the "
                                "reference's recorded uniforms through the same kernel)"},
        "phase_ms": {"collect_plus_gae": collect_ms, "update": update_ms},
        "ms_per_step_per_rank": rank_ms,
        "collective": ({"backend": dist.get_backend(), "ranks": dist.get_world_size(), "distinct_gpus": 1 if rehearsal else world,
                        "grad_allreduce_per_iteration": int(runner.alg.num_learning_epochs) * int(runner.alg.num_mini_batches),
                        "bucket_bytes": 4 * (runner.alg.bucket.numel + 8)} if (world > 1 or force_dist) else None),
    }
    if rank == 0:
        env_rate, env_step_s = time_env_path(env, T, iters=20)
        k_s = time_obs_kernel(env)
        bytes_launch = obs_kernel_bytes_per_env(env.plan) * args.num_envs
        achieved = bytes_launch / k_s / 1e9
        out["env_step_path"] = {"value": env_rate, "unit": "env-steps/s", "us_per_env_step_batch": env_step_s * 1e6,
                                "what": "imx_action_process + imx_terminations_rewards + imx_observations per step, imx_gae per 24 steps; no policy"}
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r02_pmc_traffic.json")  # PMC passes of tools/pmc_obs.py (same config), see profiles/README.md
        if os.path.exists(tf) and args.num_envs == 4096 and args.task == TASK:
            traffic = json.load(open(tf)).get("k_obs_bytes_per_launch")
        out["roofline"] = {"bound": "hbm", "kernel": "k_obs_lean<false> (observation assembly + fused height-scanner ray-cast)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "bytes_per_launch": bytes_launch, "avg_launch_us": k_s * 1e6,
                           "peak_measured_copy": measured_copy_gbs(device)}  # GB/s of a 1 GiB device-to-device copy on this box
        # the same kernel at 16x the batch (65 536 envs): what the layout reaches once launch latency is amortised
        if os.environ.get("IMX_BENCH_LARGE_N", "1") == "1" and not args.no_large_n and args.task == TASK:
            try:
                big_n = 65536
                _, env_big, _ = build_env(args.task, big_n, device, 7, 1, tuple(args.terrain_tiles), mesh=env.terrain)
                env_big.reset()
                kb = time_obs_kernel(env_big, launches=50)
                out["roofline_large_n"] = {"num_envs": big_n, "kernel": out["roofline"]["kernel"], "avg_launch_us": kb * 1e6,
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
