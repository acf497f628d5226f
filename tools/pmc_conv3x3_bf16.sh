#!/bin/bash
# PMC passes on the halo-tiled 3x3 micro-benchmark (tools/bench_conv3x3_bf16.py; counters in their own runs, no trace domains, and a
# kernel trace for the durations).  Usage: tools/pmc_conv3x3_bf16.sh [bf16|f32|bf16_colsum|wgrad_bf16|wgrad_f32]; summary: tools/pmc_conv3x3_summarize.py <dir>.
P=${1:-bf16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_c3_$P
mkdir -p $O
export TG_BENCH_ONLY=$P
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- python3 tools/bench_conv3x3_bf16.py > $O/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/b -- python3 tools/bench_conv3x3_bf16.py > $O/b.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 tools/bench_conv3x3_bf16.py > $O/t.log 2>&1
python3 tools/pmc_conv3x3_summarize.py $O | tee $O/summary.txt
