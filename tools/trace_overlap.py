"""Reads a rocprofv3 kernel trace (csv) and reports, over the span of the self-play kernels, how much of the wall time
has 0, 1, 2+ kernels in flight and which kernel classes are alone on the chip.  Usage: trace_overlap.py <dir>"""
import csv, glob, os, sys
from collections import defaultdict
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
ev = []
for f in files:
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        cls = "tree" if "tree_step" in n else "conv" if "conv_pair" in n or "conv_rows" in n or "conv_board" in n else "fc" if "fc_split" in n or "fc16_" in n else None
        if cls:
            ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), cls))
ev.sort()
# keep the last 60 % of the launches (steady state)
ev = ev[int(len(ev) * 0.4):]
pts = []
for s, e, c in ev:
    pts.append((s, 1, c)); pts.append((e, -1, c))
pts.sort()
active = defaultdict(int)
hist = defaultdict(int)
t_prev = pts[0][0]
for t, d, c in pts:
    key = tuple(sorted(k for k, v in active.items() if v > 0 for _ in range(v)))
    hist[key] += t - t_prev
    t_prev = t
    active[c] += d
tot = sum(hist.values())
for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
    print(f"{v / tot * 100:6.2f} %  in flight: {' + '.join(k) if k else '(nothing)'}")
print("launches:", len(ev), "span ms:", tot / 1e6)
idle = hist.get((), 0)
# idle gaps: how many, how long
gaps = []
end_max = ev[0][1]
for s, e, c in ev[1:]:
    if s > end_max:
        gaps.append(s - end_max)
    end_max = max(end_max, e)
gaps.sort()
import statistics
print("idle gaps:", len(gaps), "total ms:", sum(gaps) / 1e6, "median us:", statistics.median(gaps) / 1e3 if gaps else 0,
      "over 100 us:", sum(1 for g in gaps if g > 1e5), "their total ms:", sum(g for g in gaps if g > 1e5) / 1e6,
      "largest ms:", [round(g / 1e6, 2) for g in gaps[-5:]])
if gaps:
    # the largest gap is the host between two bench steps (fence, all-gather, bookkeeping), not the lock-step loop
    print(f"nothing in flight, the largest gap left out: {(idle - gaps[-1]) / (tot - gaps[-1]) * 100:.2f} % of the span")
