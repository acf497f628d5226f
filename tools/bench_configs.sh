#!/bin/bash
# Runs on the GPU box (via gpurun): tools/bench_config.py for every non-headline BASELINE configuration -> gpurun_out/configs/<name>.json,
# and the exposed-time split (tools/exposed_time.py) of the configs[3] step.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/configs
mkdir -p $O
for c in svhn-bf16 cifar10-bf16 svhn mnist stress64; do
  python3 tools/bench_config.py --config $c > $O/$c.json 2> $O/$c.err
  python3 -c "import json,sys; d=json.load(open('$O/$c.json')); print('$c', d['ms_per_step'], d.get('exec_mode_chosen'), {k: v['ms'] for k, v in d['classes'].items()})"
done
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 tools/bench_config.py --config svhn-bf16 --steps 60 --warmup 60 > $O/svhn-bf16_traced.json 2> $O/tl.err
TG_TRACE_MARK=step_inc TG_TRACE_MARK_PER=3 python3 tools/exposed_time.py $O/tl 70 110 > $O/svhn-bf16_exposed.txt     # iterations 70..110: the timed region (the execution mode is decided within the first 49)
cat $O/svhn-bf16_exposed.txt
rm -rf $O/tl
