"""Timeline of ONE steady-state PPO update from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py:
python tools/timeline.py <kernel_trace.csv>.  Prints, for the last update in the trace, the wall time, the time with no kernel
running, the time with exactly one / two or more kernels running, and the per-minibatch sequence with gaps."""
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"])) for r in rows]
k.sort()
names = [x[2] for x in k]
# an update = the kernels between the last k_gae.. of the trace that is followed by a whole update and that update's last k_adam_apply
adam = [i for i, n in enumerate(names) if n.startswith("k_adam_apply")]
gae = [i for i, n in enumerate(names) if n.startswith("k_gae") or n.startswith("k_adv_normalize")]
last_gae = max(g for g in gae if sum(1 for i in adam if i > g) >= 20)
start = last_gae + 1
nxt_obs = next((i for i in range(start, len(k)) if "k_obs" in names[i] or "k_mlp_infer" in names[i]), len(k))
end = max(i for i in adam if start < i < nxt_obs)
seg = k[start:end + 1]
t0, t1 = seg[0][0], max(x[1] for x in seg)
print(f"update: {len(seg)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, queues {sorted(set(x[3] for x in seg))}")
ev = []
for s, e, n, q in seg:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = {0: 0, 1: 0, 2: 0}
cur, prev = 0, t0
for t, d in ev:
    busy[min(cur, 2)] += t - prev
    prev = t
    cur += d
tot = t1 - t0
print("no kernel running %.3f ms (%.1f %%), one %.3f ms (%.1f %%), two or more %.3f ms (%.1f %%)" % (
    busy[0] / 1e6, 100 * busy[0] / tot, busy[1] / 1e6, 100 * busy[1] / tot, busy[2] / 1e6, 100 * busy[2] / tot))
print("sum of kernel durations %.3f ms" % (sum(e - s for s, e, _, _ in seg) / 1e6))
# one minibatch in the middle (minibatches end with k_adam_apply): kernels with start offset, duration, queue
ends = [i for i, x in enumerate(seg) if x[2].startswith("k_adam_apply")]
a, b = ends[9] + 1, ends[10]
m0 = seg[ends[9]][1]
print(f"--- minibatch 10: {(seg[b][1] - m0) / 1e3:.1f} us from the end of the previous Adam step to the end of its own")
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "")[:44]
for s, e, n, q in seg[a:b + 1]:
    print(f"  +{(s - m0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  q{q}  {short(n)}")
