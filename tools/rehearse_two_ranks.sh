#!/bin/bash
# Two ranks of bench.py on the ONE GPU of a box (collectives over gloo), started by hand so that a rank that stops
# making progress dumps its Python stack (SIGABRT + faulthandler) instead of hanging the launcher.
# Usage (GPU box, repository root): bash tools/rehearse_two_ranks.sh [seconds before the dump, default 120]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/rehearsal
limit=${1:-120}
export NNB_BENCH_REHEARSAL_ONE_GPU=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 20 "$limit" python -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 \
      > gpurun_out/rehearsal/rank$r.json 2> gpurun_out/rehearsal/rank$r.err &
done
for i in $(seq 1 $((limit / 10 + 4))); do
  sleep 10; echo "waiting $((i * 10)) s"
  if ! jobs -r | grep -q .; then break; fi
done
wait
echo "rank 0:"; cut -c1-400 gpurun_out/rehearsal/rank0.json; tail -25 gpurun_out/rehearsal/rank0.err
echo "rank 1:"; tail -25 gpurun_out/rehearsal/rank1.err
