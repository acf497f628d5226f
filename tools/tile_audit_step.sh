#!/bin/bash
# Runs on the GPU box (via gpurun): the per-launch table of one instrumented iteration of the bench workload with every tile candidate of
# the generic MFMA kernel forced in turn (TG_IGEMM_TILE: where the tile is no candidate the model's pick runs).  tools/tile_audit_step.py
# prints, per launch shape, the model's pick beside the best forced tile.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tile_audit
mkdir -p $O
TG_PROF_DUMP=$O/model.csv python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --prof-iters 2 --soak-seconds 0 --exec eager > /dev/null 2> $O/model.err
for T in 128,128 64,128 64,64 128,64 32,128 128,32; do
  TG_IGEMM_TILE=$T TG_PROF_DUMP=$O/tile_$T.csv python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --prof-iters 2 --soak-seconds 0 --exec eager > /dev/null 2> $O/tile_$T.err
  echo done $T
done
python3 tools/tile_audit_step.py $O
