#!/bin/bash
# Round-3 evidence in one gpurun call (each rocprofv3 counter set in its own run; no trace domains mixed with --pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
rm -rf $O && mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json.log 2> $O/bench_default.err
echo "bench rc=$?"
python3 bench.py --full-step --no-cpu-baseline --no-large-n > $O/bench_full_step.json.log 2> $O/bench_full_step.err
echo "bench full-step rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-large-n > $O/prof_bench.log 2>&1
echo "prof rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -o b -- python3 $R/bench.py --full-step --steps 6 --warmup 3 --no-cpu-baseline --no-large-n > $O/prof_full.log 2>&1
echo "prof full rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/pmc_obs.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/pmc_obs.py > $O/pmc_write.log 2>&1
for n in 4096 65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/pstep$n -o s -- python3 $R/tools/step_bench.py --num-envs $n > $O/step_$n.log 2>&1
done
cd $R
python tools/prof_summary.py $O/prof_bench 45 > $O/rocprof_bench_kernel_stats.txt
python tools/prof_summary.py $O/prof_full 45 > $O/rocprof_bench_full_step_kernel_stats.txt
python tools/prof_summary.py $O/pstep4096 8 > $O/rocprof_step_kernels_4096.txt
python tools/prof_summary.py $O/pstep65536 8 > $O/rocprof_step_kernels_65536.txt
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch_size.txt
python tools/pmc_summary.py $O/pmc_write > $O/pmc_write_size.txt
python tools/pmc_traffic_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null
for t in Isaac-Velocity-Flat-Anymal-C-v0 Isaac-Velocity-Rough-G1-v0 Isaac-Cartpole-v0; do python3 bench.py --task $t --no-cpu-baseline --no-large-n > $O/bench_$t.json.log 2> $O/bench_$t.err; echo "$t rc=$?"; done
for t in Isaac-Velocity-Flat-Anymal-C-v0 Isaac-Velocity-Rough-G1-v0; do python3 bench.py --task $t --full-step --steps 5 --no-cpu-baseline --no-large-n > $O/bench_full_step_$t.json.log 2> $O/bench_full_step_$t.err; echo "full-step $t rc=$?"; done
IMX_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-large-n > $O/bench_forced_dist_rccl.json.log 2> $O/bench_forced.err; echo "forced rc=$?"
IMX_REHEARSE_ONE_GPU=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-large-n > $O/bench_rehearsal_2ranks_gloo.json.log 2> $O/bench_rehearsal.err; echo "rehearsal rc=$?"
rm -rf $O/prof_bench $O/prof_full $O/pstep4096 $O/pstep65536 $O/pmc_fetch $O/pmc_write
ls $O; head -c 400 $O/bench_default.json.log; echo
