#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace of the bench command in each execution mode + tools/exposed_time.py on it.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/exposed
mkdir -p $O
for MODE in ${1:-plan graph}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/$MODE -- python3 bench.py --steps 40 --warmup 20 --no-cpu-baseline --soak-seconds 0 --prof-iters 1 --exec $MODE > $O/$MODE.json 2> $O/$MODE.err
  python3 tools/exposed_time.py $O/$MODE > $O/$MODE.txt
  cat $O/$MODE.txt
  rm -rf $O/$MODE
done
