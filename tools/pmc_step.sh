#!/bin/bash
# SQ counter passes over the env step kernels (each --pmc set in its own run; no trace domains mixed in)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_step_a $R/gpurun_out/pmc_step_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_step_a -o a -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmc_step_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/pmc_step_b -o b -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmc_step_b.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_step_a | grep "k_term_rew\|k_obs\|k_action"; python tools/pmc_summary.py gpurun_out/pmc_step_b | grep "k_term_rew\|k_obs\|k_action"
