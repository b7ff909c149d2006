#!/bin/bash
# update time against the CU budget of imx_mlp_dw (actor and critic backward passes share the chip on two streams)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
for b in 0 192 128 96 64; do
  IMX_DW_CU_BUDGET=$b python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-large-n > $O/b_$b.json 2> $O/b_$b.err
  python3 - <<PY
import json
d=json.loads(open("$O/b_$b.json").read().strip().splitlines()[-1])
print("budget $b", round(d["ms_per_step"],3), d["phase_ms"], d["config"]["update"])
PY
done
