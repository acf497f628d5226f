#!/bin/bash
# PMC passes on the bf16 3x3 micro-benchmark (tools/bench_conv3x3_bf16.py; counters in their own runs, no trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_c3
TG_BENCH_ONLY=bf16 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_c3/a -- python3 tools/bench_conv3x3_bf16.py > gpurun_out/pmc_c3/a.log 2>&1
TG_BENCH_ONLY=bf16 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmc_c3/b -- python3 tools/bench_conv3x3_bf16.py > gpurun_out/pmc_c3/b.log 2>&1
find gpurun_out/pmc_c3 -name "*counter_collection.csv" | head
