"""PPO update time on one GPU with (1) no process group use, (2) torch.distributed all-reduce + div_, (3) the reduce skipped but the
process group alive -- same runner, same data, a single-rank nccl group.  PG_FIRST=1 creates the process group before the runner
(the order of a real launch): RCCL's streams then take hardware queues first and, with ROCm's default of 4 hardware queues per
process, two of the update's three streams share one -- 20.2 ms instead of 17.7 ms per update.  GPU_MAX_HW_QUEUES=8 (set by
bench.py / isaaclab_amd before HIP starts; IN_SCRIPT=1 here does the same) removes the sharing."""
import os
import sys

if os.environ.get("IN_SCRIPT") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29545")
import torch
import torch.distributed as dist

from bench import build_env
from isaaclab_amd.rsl_rl import OnPolicyRunner, RslRlVecEnvWrapper

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
PG_FIRST = os.environ.get("PG_FIRST") == "1"
if PG_FIRST:  # the order bench.py / a real launch uses: process group before any of the update's streams exist
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    tmp = torch.zeros(8, device=dev)
    dist.broadcast(tmp, src=0)
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
venv = RslRlVecEnvWrapper(env, clip_actions=fx["agent"].get("clip_actions"))
runner = OnPolicyRunner(venv, fx["agent"], log_dir=None, device=str(dev), use_graph=True)
runner.train_mode()
alg = runner.alg


def iteration():
    runner.collect()
    with torch.inference_mode():
        alg.compute_returns(runner.last_obs)
    alg.update()


def timed(tag, n=6):
    for _ in range(2):
        iteration()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        runner.collect()
        with torch.inference_mode():
            alg.compute_returns(runner.last_obs)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        alg.update()
        b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    print(f"{tag:50s} update {tot / n:7.3f} ms", flush=True)


timed("no process group")
if not PG_FIRST:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
timed("process group alive, no reduce")
alg.is_multi_gpu, alg.gpu_world_size = True, 1
timed("torch.distributed all_reduce + div_")
real = alg.reduce_parameters
alg.reduce_parameters = lambda: None
timed("is_multi_gpu, reduce skipped")
dist.destroy_process_group()
