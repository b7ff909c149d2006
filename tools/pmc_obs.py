"""Workload for the PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, one counter set per
pass): 20 launches of each env kernel at the bench configuration + a calibration copy of known size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build_env

dev = torch.device("cuda:0")
fx, env, ntri = build_env("Isaac-Velocity-Rough-Anymal-C-v0", 4096, dev, 42, 4, (10, 20))
env.reset()
a = torch.zeros(4096, 12, device=dev)
for _ in range(20):
    env.step(a)
# calibration: 256 MiB read + 256 MiB write, streaming float4
src = torch.empty(64 * 1024 * 1024, device=dev).normal_()
dst = torch.empty_like(src)
for _ in range(5):
    dst.copy_(src)
torch.cuda.synchronize()
print("done")
