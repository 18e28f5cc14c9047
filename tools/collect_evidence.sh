#!/bin/bash
# Collects, on one GPU box, the measurements the design notes lean on, as text files under gpurun_out/evidence/
# (copied into profiles/ afterwards).  Usage (repository root): bash tools/collect_evidence.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/evidence
rm -rf "$out"; mkdir -p "$out"
# per-kernel times, per-phase cycle stamps, clocks and the linear layer's ablations
timeout -k 10 120 tools/split_bench 3979 0 > "$out/split_bench.txt" 2>&1 || exit 1
timeout -k 10 120 tools/split_bench 3979 1 > "$out/split_bench_all_heads.txt" 2>&1 || exit 1
# what runs beside what on the two chains
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-lines > "$out/trace_bench.json" 2> "$out/trace_bench.err" || exit 1
python3 tools/trace_overlap.py "$out/trace" > "$out/overlap.txt" 2>&1 || exit 1
rm -rf "$out/trace"
# the linear layer's fabric reads against the row count
bash tools/fc_fetch_by_rows.sh > "$out/fc_fetch_by_rows.txt" 2>&1 || exit 1
# the multi-rank code path of bench.py, two ranks on the one GPU of this box (collectives over gloo)
NNB_BENCH_REHEARSAL_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 > "$out/bench_two_ranks_one_gpu.json" 2> "$out/bench_two_ranks_one_gpu.err" || exit 1
# one rank under the launcher: the collective goes over RCCL
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus 1 --steps 2 --warmup 1 --no-extra-lines --no-cpu-baseline > "$out/bench_one_rank_rccl.json" 2> "$out/bench_one_rank_rccl.err" || exit 1
echo collected
