#!/bin/bash
# Times the headline stream with each library variant of tools/build_variants.py on ONE box (same clocks for all).
# Usage (GPU box, repository root): bash tools/sweep_knobs.sh > gpurun_out/sweep.txt
cd "$(dirname "$0")/.."
run() {
  local name=$1 lib=$2
  NNB_LIB=$lib timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-extra-lines --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['kernel_ms']
print(f\"$name: {d['value']:.0f} games/s, {d['sims_per_s'] / 1e6:.2f} M sims/s, {d['ms_per_step']:.0f} ms/step, rounds {d['stats']['lockstep_rounds']}, fill {d['stats']['leaf_batch_fill']:.3f}, tree {k['tree_step'] * 1e3:.0f} us conv {k['conv2'] * 1e3:.0f} us fc {k['fc'] * 1e3:.0f} us\")" || exit 1
}
run "product (6 terminal sims per step, look every 64 rounds)" alphazero_amd/libnnb.so || exit 1
for v in ts3 ts10 look32 look128 look256; do run "$v" tools/variants/libnnb_$v.so || exit 1; done
run "product again" alphazero_amd/libnnb.so
