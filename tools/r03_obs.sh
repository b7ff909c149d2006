#!/bin/bash
# k_obs_lean under rocprofv3 at 4096 and 65536 envs (tools/step_bench.py) + the parity tests that guard the ray path
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03obs; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_env_gpu.py -m gpu -x -q -k "ray or goldens or full_size or boundaries or kitchen" > $O/t.log 2>&1; tail -2 $O/t.log
cd /tmp
for n in 4096 65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$n -o s -- python3 $R/tools/step_bench.py --num-envs $n > $O/step_$n.log 2>&1
  (cd $R && python tools/prof_summary.py $O/p$n 6 | grep -v "^#" > $O/stats_$n.txt); rm -rf $O/p$n
  cat $O/stats_$n.txt | cut -c1-60,80-140
done
