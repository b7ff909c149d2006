"""Data-parallel PPO.update with two ranks on the one GPU of the test box (gloo group, both ranks on cuda:0): every kernel of the
update is the HIP path, only the transport differs from RCCL.  Replicas must stay identical although their rollouts differ."""
import os
import socket
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import isaaclab_amd  # noqa: F401  (hardware-queue setting before HIP starts)
    from isaaclab_amd.rsl_rl.actor_critic import ActorCritic
    from isaaclab_amd.rsl_rl.ppo import PPO

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, N, D, A = 8, 64, 48, 12
    torch.manual_seed(100 + rank)  # different initial weights per rank: broadcast_parameters must fix that
    pol = ActorCritic(D, D, A, actor_hidden_dims=[128, 64], critic_hidden_dims=[128, 64], init_noise_std=1.0)
    alg = PPO(pol, device="cuda:0", multi_gpu_cfg={"global_rank": rank, "local_rank": rank, "world_size": world}, num_learning_epochs=2,
              num_mini_batches=4, schedule="adaptive", desired_kl=0.01, learning_rate=1e-3, entropy_coef=0.005, max_grad_norm=1.0,
              clip_param=0.2, value_loss_coef=1.0, use_clipped_value_loss=True)
    alg.update_graph = False
    alg.init_storage("rl", N, T, (D,), (0,), (A,))
    alg.broadcast_parameters()
    g = torch.Generator().manual_seed(500 + rank)  # each rank its own rollout
    for it in range(2):
        st = alg.storage
        st.observations.copy_(torch.randn(T, N, D, generator=g))
        with torch.no_grad():
            mu = alg.policy.actor(st.observations.flatten(0, 1)).view(T, N, A)
            val = alg.policy.critic(st.observations.flatten(0, 1)).view(T, N, 1)
        sigma = alg.policy.std.detach().expand(T, N, A).contiguous()
        act = mu + sigma * torch.randn(T, N, A, generator=g).cuda()
        st.mu.copy_(mu); st.sigma.copy_(sigma); st.actions.copy_(act); st.values.copy_(val)
        st.actions_log_prob.copy_(torch.distributions.Normal(mu, sigma).log_prob(act).sum(-1, keepdim=True))
        st.returns.copy_(val + 0.3 * torch.randn(T, N, 1, generator=g).cuda())
        st.advantages.copy_(torch.randn(T, N, 1, generator=g))
        st.step = T
        alg.update()
    torch.cuda.synchronize()
    flat = alg.bucket.flat.detach().cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    lrs = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(lrs, torch.tensor([alg.learning_rate]))
    ok = all(torch.equal(x, gathered[0]) for x in gathered) and all(float(x) == float(lrs[0]) for x in lrs) and bool(torch.isfinite(flat).all())
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_keep_identical_replicas():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(0) is True and out.get(1) is True


@pytest.mark.gpu
def test_bench_two_rank_rehearsal():
    """The driver's own form ``python bench.py --gpus 2 ...`` on the one GPU of the test box: the parent (which never initialises HIP)
    starts two ranks through torch.distributed.run, both on cuda:0 over a gloo group (IMX_REHEARSE_ONE_GPU=1), and relays rank 0's
    single JSON line."""
    import json
    import subprocess

    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["IMX_REHEARSE_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--num-envs", "512",
                        "--terrain-tiles", "2", "3", "--no-cpu-baseline", "--no-large-n"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and len(out["ms_per_step_per_rank"]) == 2
    c = out["collective"]
    assert c["ranks"] == 2 and c["backend"] == "gloo" and c["distinct_gpus"] == 1 and len(c["gpu_pci_ids"]) == 2
    # the self-diagnosis fields: all-reduce time from HIP events (2 steps x 5 epochs x 4 minibatches), per-rank set-up times
    assert c["allreduce_us"]["count"] == 2 * c["grad_allreduce_per_iteration"] and c["allreduce_us"]["mean"] > 0
    pr = out["per_rank"]
    assert all(len(pr[k]) == 2 for k in ("terrain_build_s", "graph_capture_s", "allreduce_us_mean")) and min(pr["graph_capture_s"]) > 0
    assert out["value"] > 0


@pytest.mark.gpu
def test_bench_refuses_a_rank_count_that_does_not_match_gpus():
    """A launcher that started a different number of ranks than --gpus says must end in a clear non-zero exit, not a hang or a wrong line."""
    import subprocess

    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--num-envs", "256",
                        "--no-cpu-baseline", "--no-large-n"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
