#!/bin/bash
# SQ counter passes over the env step kernels at 65 536 envs (each --pmc set in its own run; no trace domains mixed in)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export IMX_PMC_N=65536
rm -rf $R/gpurun_out/pmcL_a $R/gpurun_out/pmcL_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcL_a -o a -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcL_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmcL_b -o b -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcL_b.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN TCP_TCC_READ_REQ_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmcL_c -o c -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcL_c.log 2>&1
cd $R && (python tools/pmc_summary.py gpurun_out/pmcL_a; python tools/pmc_summary.py gpurun_out/pmcL_b; python tools/pmc_summary.py gpurun_out/pmcL_c) | grep "k_term_rew\|k_obs\|k_action" > gpurun_out/pmc_step_sq_65536.txt
rm -rf $R/gpurun_out/pmcL_*/*.db
cat gpurun_out/pmc_step_sq_65536.txt
