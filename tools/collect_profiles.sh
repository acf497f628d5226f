#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel stats of the same command, per-launch table.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r01}
O=gpurun_out/profiles_$R
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
TG_PROF_DUMP=$O/launches.csv python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --prof-iters 1 --soak-seconds 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rp -- python3 bench.py --no-cpu-baseline --soak-seconds 0 > $O/rp_bench.json 2> $O/rp.err
cp $O/rp/*/*kernel_stats.csv $O/kernel_stats.csv
if [ -z "$SKIP_SERIAL" ]; then
# the same workload as ONE chain (--exec graph): per-kernel durations without the two-stream mode's concurrent kernels sharing the chip
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rps -- python3 bench.py --no-cpu-baseline --soak-seconds 0 --exec graph > $O/rps_bench.json 2> $O/rps.err
cp $O/rps/*/*kernel_stats.csv $O/kernel_stats_serial.csv
fi
tail -c 400 $O/bench.json
