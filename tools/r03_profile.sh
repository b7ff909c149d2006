#!/bin/bash
# Round-3 evidence in one gpurun call (each rocprofv3 counter set in its own run; no trace domains mixed with --pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
rm -rf $O && mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json.log 2> $O/bench_default.err
echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-large-n > $O/prof_bench.log 2>&1
echo "prof rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/pmc_obs.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/pmc_obs.py > $O/pmc_write.log 2>&1
cd $R
python tools/prof_summary.py $O/prof_bench 45 > $O/rocprof_bench_kernel_stats.txt
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch_size.txt
python tools/pmc_summary.py $O/pmc_write > $O/pmc_write_size.txt
python tools/pmc_traffic_json.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > /dev/null
rm -rf $O/prof_bench/*trace.csv $O/prof_bench/*/*trace.csv $O/pmc_*/*.db $O/pmc_*/*/*.db
ls $O; head -c 600 $O/bench_default.json.log; echo; head -30 $O/rocprof_bench_kernel_stats.txt
