#!/bin/bash
# HBM traffic of the dominant kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), no trace
# domains, on the same bench command.  Output: gpurun_out/traffic/{fetch,write}/.../*counter_collection.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/traffic
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --prof-iters 1 --no-graph --soak-seconds 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --prof-iters 1 --no-graph --soak-seconds 0 > $O/write.log 2>&1
ls $O/fetch/*/ $O/write/*/
