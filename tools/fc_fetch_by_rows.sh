#!/bin/bash
# Fabric-side read bytes of the float32-faithful linear layer as a function of the row count.  The kernel uses row
# tiles of 32*t rows, t = ceil(rows / 1024); XCDs 0-3 take the first two 256-column blocks, XCDs 4-7 the other two, and
# XCD x & 3 takes row blocks 8*(x & 3) .. +7 of the 32 a group of 256 blocks covers.  So the row count decides how many
# XCDs take part; each of them fetches half the weight matrix once, and every activation row is fetched by two XCDs
# (plus, per row, the 2 KB of partial sums one K half hands to the other).  One FETCH_SIZE pass per size.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/fc_fetch
rm -rf "$out"; mkdir -p "$out"
for n in 512 1024 2048 4096; do
  timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/n$n" -o run -- python3 tools/net_latency.py $n > "$out/n$n.log" 2>&1 || exit 1
  python3 - "$out/n$n" $n <<'PY'
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
vals = [float(r["Counter_Value"]) for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and ("fc_split_kernel" in r["Kernel_Name"] or "fc16_kernel" in r["Kernel_Name"])]
act, w = n * 46080 / 1e6, 47.2
t = (n + 1023) // 1024
row_blocks = (n + 32 * t - 1) // (32 * t)
xcds = 2 * min(4, (row_blocks + 7) // 8)
hand = n * 2048 / 1e6
print(f"rows {n}: {len(vals)} launches, FETCH_SIZE x 2 = {2 * sum(vals) / len(vals) * 1024 / 1e6:.1f} MB per launch; model: "
      f"activations 2 x {act:.1f} MB + weights {w / 2} MB x {xcds} XCDs + hand-off {hand:.1f} MB = {2 * act + w / 2 * xcds + hand:.1f} MB")
PY
  rm -rf "$out/n$n"
done
