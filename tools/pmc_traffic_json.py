#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the two PMC passes of tools/pmc_obs.py (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs):
per-launch HBM bytes of the env kernels with the gfx950 read correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE is in
KiB and counts half of the bytes of 16-B-per-lane reads: doubled; checked on the calibration copies of the same run).

    python tools/pmc_traffic_json.py <fetch_dir> <write_dir> <out.json> [round tag]
"""
import collections
import csv
import glob
import json
import sys


def means(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = means(sys.argv[1]), means(sys.argv[2])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/r03_profile.sh) -- python3 tools/pmc_obs.py; "
                 "Isaac-Velocity-Rough-Anymal-C-v0, 4096 envs, 1.20 M-triangle terrain, 4 state snapshots; means over the launches of 20 env steps",
       "units": "FETCH_SIZE / WRITE_SIZE are KiB; gfx950 correction: FETCH_SIZE x2 (calibration copies of the same run listed below)"}
for key, pat in (("k_obs", "k_obs"), ("k_term_rew", "k_term_rew"), ("k_action", "k_action")):
    f = [v for (n, c), v in fetch.items() if pat in n and c == "FETCH_SIZE"]
    w = [v for (n, c), v in write.items() if pat in n and c == "WRITE_SIZE"]
    if f and w:
        out[key] = {"FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0]}
        out[key + "_bytes_per_launch"] = int((2.0 * f[0] + w[0]) * 1024)
        out[key + "_uncorrected_bytes_per_launch"] = int((f[0] + w[0]) * 1024)
cal_f = [v for (n, c), v in fetch.items() if "copyBuffer" in n and c == "FETCH_SIZE"]
cal_w = [v for (n, c), v in write.items() if "copyBuffer" in n and c == "WRITE_SIZE"]
out["calibration_copy_KiB"] = {"FETCH_SIZE_mean_over_copy_kernels": cal_f, "WRITE_SIZE_mean_over_copy_kernels": cal_w}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
