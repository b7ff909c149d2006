#!/bin/bash
# per-kernel device durations of the env step kernels: rocprofv3 kernel trace of tools/step_bench.py (run through gpurun)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -o step -- python3 $GRAFT_REPO_ROOT/tools/step_bench.py --num-envs ${1:-4096} > $GRAFT_REPO_ROOT/gpurun_out/prof_step.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/prof_summary.py gpurun_out/prof_step 12
