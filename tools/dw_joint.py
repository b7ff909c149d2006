"""First-layer dW of actor and critic: two imx_mlp_dw launches (N = 512 each) on two streams at once against one launch over the stacked
outputs (N = 1024), same X."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from isaaclab_amd._lib import check, lib

dev = torch.device("cuda:0")
L = lib()
M, H, K, Kp = 24576, 512, 235, 236
X = torch.randn(M, Kp, device=dev)
dY = torch.randn(M, 2 * H, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
nb = int(L.imx_mlp_scratch_bytes(M, 2 * H, K))
scr = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
dW, db = torch.empty(2 * H, K, device=dev), torch.empty(2 * H, device=dev)


def pair():
    for k, st in enumerate((s1, s2)):
        with torch.cuda.stream(st):
            check(L.imx_mlp_dw(M, H, K, dY[:, k * H:].data_ptr(), 2 * H, X.data_ptr(), Kp, dW[k * H:].data_ptr(), db[k * H:].data_ptr(),
                               scr[k].data_ptr(), nb, st.cuda_stream))


def joint():
    with torch.cuda.stream(s1):
        check(L.imx_mlp_dw(M, 2 * H, K, dY.data_ptr(), 2 * H, X.data_ptr(), Kp, dW.data_ptr(), db.data_ptr(), scr[0].data_ptr(), nb, s1.cuda_stream))


for fn, tag in ((pair, "two launches, two streams"), (joint, "one stacked launch")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s1)
    for _ in range(30):
        fn()
    s1.wait_stream(s2)
    b.record(s1)
    torch.cuda.synchronize()
    print(f"{tag:28s} {a.elapsed_time(b) * 1e3 / 30:7.1f} us")
    out = dW.clone()
    print("   checksum", float(out.double().sum()))
