#!/bin/bash
# Round-2 evidence, collected in one gpurun call (each rocprofv3 counter set in its own run; no trace domains mixed with --pmc).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
rm -rf $O && mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json.log 2> $O/bench_default.err
echo "bench rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-large-n > $O/prof_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/pmc_obs.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/pmc_obs.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq_a -o a -- python3 $R/tools/pmc_step.py > $O/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_sq_b -o b -- python3 $R/tools/pmc_step.py > $O/pmc_sq_b.log 2>&1
cd $R
python tools/prof_summary.py $O/prof_bench 40 > $O/rocprof_bench_kernel_stats.txt
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_fetch_size.txt
python tools/pmc_summary.py $O/pmc_write > $O/pmc_write_size.txt
(python tools/pmc_summary.py $O/pmc_sq_a; python tools/pmc_summary.py $O/pmc_sq_b) | grep "k_obs\|k_term_rew\|k_action" > $O/pmc_step_sq.txt
bash tools/prof_step.sh 4096 > $O/step_kernels_4096.txt 2>&1
bash tools/prof_step.sh 65536 > $O/step_kernels_65536.txt 2>&1
IMX_REHEARSE_ONE_GPU=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-large-n > $O/bench_rehearsal_2ranks_gloo.json.log 2> $O/bench_rehearsal.err
IMX_FORCE_DIST=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-large-n > $O/bench_forced_dist_rccl.json.log 2> $O/bench_forced.err
rm -rf $O/prof_bench/*trace.csv $O/pmc_*/*.db
ls $O; cat $O/bench_default.json.log | head -c 1500
