#!/bin/bash
# Collect SQ counters for the conv kernels (two separate --pmc passes; no trace domains mixed in).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc2/p1 -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > gpurun_out/pmc2/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_IFETCH --output-format csv -d gpurun_out/pmc2/p2 -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > gpurun_out/pmc2/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc2/p3 -- python bench.py --steps 2 --warmup 0 --no-profile --cpu-budget 0 > gpurun_out/pmc2/p3.log 2>&1
ls gpurun_out/pmc2/*/*/ | head
