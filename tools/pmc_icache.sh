#!/bin/bash
# instruction-cache counters of the env step kernels (one --pmc pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_ic -o i -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmc_ic.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_ic | grep "k_term_rew\|k_obs\|k_action"
rm -rf $R/gpurun_out/pmc_ic/*.db
tail -3 $R/gpurun_out/pmc_ic.log
