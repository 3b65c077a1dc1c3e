cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SWEEP_STEPS=3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_a -- python3 green-marl_amd/tools/cold_sweep.py 26 4 0 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_b -- python3 green-marl_amd/tools/cold_sweep.py 26 4 0 > gpurun_out/pmc_b.log 2>&1
ls gpurun_out/pmc_a/* gpurun_out/pmc_b/*
