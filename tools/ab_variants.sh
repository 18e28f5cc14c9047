#!/bin/bash
# Times the headline stream with two or more libraries on ONE box, alternating (same clocks for all).
# Usage (GPU box, repository root): bash tools/ab_variants.sh name=lib.so name=lib.so ... > gpurun_out/ab.txt
cd "$(dirname "$0")/.."
run() {
  NNB_LIB=$2 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-lines 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernel_ms']; r = d.get('roofline', {}).get('kernel_ms') or {'conv2': 0, 'fc': 0, 'tree_step': 0}
print(f\"$1: {d['value']:.0f} games/s, {d['ms_per_step']:.0f} ms/step, in two chains tree {k['tree_step']*1e3:.0f} conv {k['conv2']*1e3:.0f} fc {k['fc']*1e3:.0f} us; one chain tree {r['tree_step']*1e3:.0f} conv {r['conv2']*1e3:.0f} fc {r['fc']*1e3:.0f} us\", flush=True)" || exit 1
}
for rep in 1 2; do
  for nv in "$@"; do run "${nv%%=*}" "${nv#*=}" || exit 1; done
done
