#!/bin/bash
# PMC passes on tools/bench_step_shapes.py (the 61 launch shapes of the CIFAR-10 step on the generic MFMA kernels): counters in their own
# runs, no trace domains; then a kernel trace for the durations.  Summary: python3 tools/pmc_igemm_summarize.py gpurun_out/pmc_shapes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_shapes
mkdir -p $O
export TG_SHAPES_ITERS=4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/a -- python3 tools/bench_step_shapes.py f32 > $O/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -- python3 tools/bench_step_shapes.py f32 > $O/b.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 tools/bench_step_shapes.py f32 > $O/t.log 2>&1
python3 tools/pmc_igemm_summarize.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
