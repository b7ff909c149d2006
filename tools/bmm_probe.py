"""Layers 1-2 of actor and critic: two library GEMMs on two streams against one batch-2 strided GEMM (torch.baddbmm)."""
import torch

dev = torch.device("cuda:0")
M = 24576
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for K, N in ((512, 256), (256, 128)):
    X = torch.randn(2, M, K, device=dev)
    W = torch.randn(2, N, K, device=dev)
    b = torch.randn(2, 1, N, device=dev)

    def pair():
        with torch.cuda.stream(s1):
            torch.addmm(b[0, 0], X[0], W[0].t())
        with torch.cuda.stream(s2):
            torch.addmm(b[1, 0], X[1], W[1].t())

    def batched():
        with torch.cuda.stream(s1):
            torch.baddbmm(b, X, W.transpose(1, 2))

    for fn, tag in ((pair, "two addmm, two streams"), (batched, "one baddbmm (batch 2)")):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s1)
        for _ in range(30):
            fn()
        s1.wait_stream(s2)
        e.record(s1)
        torch.cuda.synchronize()
        us = a.elapsed_time(e) * 1e3 / 30
        print(f"{K}->{N} {tag:26s} {us:7.1f} us  {2 * 2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s")
