"""Where do the ~137 us per gradient all-reduce go (single-rank group on one GPU)?  Times, per call and between HIP events:
(a) torch.distributed.all_reduce on the 2.29 MB bucket, (b) the same + div_, each interleaved with a small kernel on the same
stream so that the hand-off cost to and from the process group's stream shows.  Result (MI355X, one rank): 8 us and 12 us per call --
the all-reduce is NOT where a distributed run loses time; the hardware-queue count is (tools/dist_overhead.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 571801 + 8
g = torch.randn(n, device=dev)
w = torch.randn(n, device=dev)


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def base():
    w.add_(g, alpha=1e-9)


def torch_ar():
    w.add_(g, alpha=1e-9)
    dist.all_reduce(g)


def torch_ar_div():
    w.add_(g, alpha=1e-9)
    dist.all_reduce(g)
    g.div_(1.0)


print(f"small kernel alone                 {timed(base):7.1f} us")
print(f"+ torch all_reduce                 {timed(torch_ar):7.1f} us")
print(f"+ torch all_reduce + div_          {timed(torch_ar_div):7.1f} us")

dist.destroy_process_group()
