#!/bin/bash
# every random sweep once, on the committed code (summaries -> profiles/r03_fuzz_*.txt by hand); two gpurun calls: "a" and "b"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fuzz_final; mkdir -p $O; cd $R
if [ "$1" = "a" ]; then
  timeout -k 10 1000 python tools/fuzz_parity.py 1500 100 > $O/parity.txt 2>&1; echo "parity rc=$? $(tail -1 $O/parity.txt)"
else
  timeout -k 10 300 python tools/fuzz_cfg.py 500 6000 > $O/cfg.txt 2>&1; echo "cfg rc=$? $(tail -1 $O/cfg.txt)"
  timeout -k 10 300 python tools/fuzz_rollout.py 400 1000 > $O/rollout.txt 2>&1; echo "rollout rc=$? $(tail -1 $O/rollout.txt)"
  timeout -k 10 400 python tools/fuzz_kernels.py 300 77 > $O/kernels.txt 2>&1; echo "kernels rc=$?"; grep -E "cases agree|FAIL" $O/kernels.txt
  timeout -k 10 300 python tools/fuzz_orchestration.py 300 1000 > $O/orchestration.txt 2>&1; echo "orchestration rc=$? $(tail -1 $O/orchestration.txt)"
  timeout -k 10 300 python tools/fuzz_raycast.py 250 1000 > $O/raycast.txt 2>&1; echo "raycast rc=$? $(tail -1 $O/raycast.txt)"
  timeout -k 10 200 python tools/fuzz_producers.py 150 0 > $O/producers.txt 2>&1; echo "producers rc=$?"; grep -E "cases agree|FAIL" $O/producers.txt
fi
