#!/bin/bash
# Rehearsal of the N = 2 bench line on a ONE-GPU box: two ranks share device 0 and exchange over gloo (RCCL cannot put two ranks on one
# device).  Everything else is the production bench path: per-rank batches, execution-mode decision across ranks, self-test, replica
# checksums, exposed exchange time, one JSON line from rank 0.  The throughput it prints is meaningless (two replicas on one GPU).
cd $GRAFT_REPO_ROOT
PORT=29617
for R in 0 1; do
  RANK=$R LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT TG_DIST_BACKEND=gloo TG_DEVICE_INDEX=0 \
    timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 3 --soak-seconds 0 > gpurun_out/n2_rank$R.json 2> gpurun_out/n2_rank$R.err &
  PIDS[$R]=$!
done
RC=0
for R in 0 1; do wait ${PIDS[$R]} || RC=1; done
echo "rc $RC"; cat gpurun_out/n2_rank0.json; echo "--- rank 1 stdout (must be empty):"; cat gpurun_out/n2_rank1.json; tail -n 3 gpurun_out/n2_rank0.err gpurun_out/n2_rank1.err
exit $RC
