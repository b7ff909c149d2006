#!/bin/bash
# final round-2 bench lines (after the last kernel change): default run, forced single-rank RCCL run, two-rank gloo rehearsal
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02f; rm -rf $O; mkdir -p $O; cd $R
python3 bench.py > $O/bench_default.json.log 2> $O/bench_default.err; echo "default rc=$?"
IMX_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-large-n > $O/bench_forced_dist_rccl.json.log 2> $O/bench_forced.err; echo "forced rc=$?"
IMX_REHEARSE_ONE_GPU=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-large-n > $O/bench_rehearsal_2ranks_gloo.json.log 2> $O/bench_rehearsal.err; echo "rehearsal rc=$?"
for t in Isaac-Velocity-Flat-Anymal-C-v0 Isaac-Velocity-Rough-G1-v0 Isaac-Cartpole-v0; do python3 bench.py --task $t --no-cpu-baseline --no-large-n > $O/bench_$t.json.log 2> $O/bench_$t.err; echo "$t rc=$?"; done
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02f/*.json.log")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), round(d["value"]), round(d["ms_per_step"], 2), d["phase_ms"], d["config"]["update"], d.get("collective"))
    except Exception as e:
        print(f, "ERR", e)
PY
