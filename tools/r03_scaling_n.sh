#!/bin/bash
# one GPU, the headline task at other env counts per GPU (the driver's line is 4096): PPO iteration and post-physics path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
for n in 1024 4096 16384 65536; do
  python3 bench.py --num-envs $n --steps 5 --warmup 2 --no-cpu-baseline --no-large-n > $O/n_$n.json 2> $O/n_$n.err
  python3 - <<PY
import json
d=json.loads(open("$O/n_$n.json").read().strip().splitlines()[-1])
s=d["roofline_step"]
print("N=$n", round(d["value"]), "env-steps/s", round(d["ms_per_step"],2), "ms", {k:round(v,2) for k,v in d["phase_ms"].items()}, "env_step_path us", round(d["env_step_path"]["us_per_env_step_batch"],1),
      {k.split("(")[0][:18]: (round(v["avg_launch_us"],1), round(v["frac"],3)) for k,v in s.items()})
PY
done
