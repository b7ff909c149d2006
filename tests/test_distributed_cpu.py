"""CPU, gloo, world_size 2: the only data-path exchange of the multi-GPU design -- the flat gradient bucket
all-reduce with the KL slot, the parameter broadcast, and the rank-identical adaptive learning rate."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import FlatParams, adaptive_lr_, allreduce_mean_

    torch.manual_seed(100 + rank)  # different init per rank, as with seed += local_rank (train.py:119-126)
    policy = ActorCritic(8, 8, 3, actor_hidden_dims=[16, 8], critic_hidden_dims=[16, 8])
    bucket = FlatParams(policy)
    # parameters are views of the flat bucket
    n = sum(p.numel() for p in policy.parameters())
    assert bucket.numel == n and bucket.grad.numel() == n + 1
    before = bucket.flat.clone()
    dist.broadcast(bucket.flat, src=0)  # PPO.broadcast_parameters
    gathered = [torch.empty_like(bucket.flat) for _ in range(world)]
    dist.all_gather(gathered, bucket.flat)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    w = policy.actor[0].weight
    off = (w.data_ptr() - bucket.flat.data_ptr()) // 4
    assert 0 <= off < n and torch.equal(w.reshape(-1), bucket.flat[off:off + w.numel()])  # a view, not a copy
    if rank != 0:
        assert not torch.equal(before, bucket.flat)
    # local backward on a rank-specific batch writes straight into the flat gradient bucket
    torch.manual_seed(7 + rank)
    x = torch.randn(32, 8)
    loss = policy.actor(x).square().mean() + policy.critic(x).mean() + policy.std.sum()
    bucket.zero_grad()
    loss.backward()
    local = bucket.grad.clone()
    goff = (w.grad.data_ptr() - bucket.grad.data_ptr()) // 4
    assert goff == off and local[:n].abs().sum() > 0  # autograd accumulated in place into the bucket
    kl_local = 0.004 + 0.03 * rank
    bucket.grad[-1] = kl_local
    local[-1] = kl_local
    allreduce_mean_(bucket.grad, world)
    all_local = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(all_local, local)
    expect = torch.stack(all_local).sum(0) / world
    assert torch.allclose(bucket.grad, expect, rtol=0, atol=1e-7)
    # adaptive LR from the reduced KL: same decision on every rank
    lr = torch.full((1,), 1e-3)
    adaptive_lr_(lr, bucket.grad[-1], desired_kl=0.01)  # mean KL = 0.019 -> inside [0.005, 0.02] -> unchanged
    lrs = [torch.empty(1) for _ in range(world)]
    dist.all_gather(lrs, lr)
    assert all(torch.equal(v, lrs[0]) for v in lrs) and abs(float(lr) - 1e-3) < 1e-9
    adaptive_lr_(lr, torch.tensor(0.05), 0.01)
    assert abs(float(lr) - 1e-3 / 1.5) < 1e-9
    adaptive_lr_(lr, torch.tensor(0.001), 0.01)
    assert abs(float(lr) - 1e-3) < 1e-9
    adaptive_lr_(lr, torch.tensor(0.0), 0.01)  # kl == 0 never raises the LR
    assert abs(float(lr) - 1e-3) < 1e-9
    out.put((rank, float(bucket.grad[-1])))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = dict(q.get() for _ in range(2))
    assert abs(res[0] - 0.019) < 1e-6 and res[0] == res[1]


@pytest.mark.timeout(300)
def test_bench_gpus_n_launches_its_own_ranks_and_propagates_failure():
    """``python bench.py --gpus 2`` outside a launcher must start the two ranks itself (child torch.distributed.run, 127.0.0.1
    rendezvous) and exit with the child's code.  No GPU here: the ranks refuse to run ('needs an MI355X'), so the parent must print NO
    result line and return non-zero -- never a fabricated line."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("GPU present: the positive path is covered by test_bench_two_rank_rehearsal (gpu)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert r.stderr.count("needs an MI355X") >= 1 and "torch.distributed" in r.stderr + "torch.distributed"  # the ranks were started and refused
