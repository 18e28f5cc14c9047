#!/bin/bash
# The lines the design notes quote, on ONE box: default bench line, one chain, other configurations, harness.
# Usage (GPU box, repository root): bash tools/round_end_lines.sh   -> gpurun_out/final/
cd "$(dirname "$0")/.."
out=gpurun_out/final
mkdir -p "$out"
timeout -k 10 400 python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err" || exit 1
timeout -k 10 200 python bench.py --one-stream --no-extra-lines --no-cpu-baseline > "$out/bench_one_chain.json" 2> "$out/bench_one_chain.err" || exit 1
timeout -k 10 500 python tools/bench_configs.py tree az arena > "$out/other_configs.txt" 2> "$out/other_configs.err" || exit 1
timeout -k 10 120 tools/split_bench 3979 0 > "$out/split_bench.txt" 2>&1 || exit 1
echo done
