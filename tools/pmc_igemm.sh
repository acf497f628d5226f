#!/bin/bash
# PMC passes on the implicit-GEMM micro-benchmark (counters in their own runs, no trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
FWD_ONLY=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc/a -- python3 tools/bench_igemm.py 250 f32 > gpurun_out/pmc/a.log 2>&1
FWD_ONLY=1 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc/b -- python3 tools/bench_igemm.py 250 f32 > gpurun_out/pmc/b.log 2>&1
FWD_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc/t -- python3 tools/bench_igemm.py 250 f32 > gpurun_out/pmc/t.log 2>&1
ls -R gpurun_out/pmc | head -30
