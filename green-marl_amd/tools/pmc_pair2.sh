# second set of SQ counters for the binned PageRank kernels (instruction fetch, VMEM / LDS queues); see pmc_pair.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SWEEP_STEPS=3
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD --output-format csv -d gpurun_out/pmc_a -- python3 green-marl_amd/tools/cold_sweep.py 26 4 0 > gpurun_out/pmc_a.log 2>&1
rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d gpurun_out/pmc_b -- python3 green-marl_amd/tools/cold_sweep.py 26 4 0 > gpurun_out/pmc_b.log 2>&1
