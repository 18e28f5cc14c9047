"""Folds rocprofv3 output into the per-kernel summaries kept under profiles/.

  python tools/summarize_pmc.py --fetch <dir> --write <dir> --stats <dir> --dtype f16 --tag r01_f16
      [--sq <dir>] [--l2 <dir>] [--outdir <dir>] [--bench-log <file>]

<dir>s are rocprofv3 -d output directories: --fetch from a `--pmc FETCH_SIZE` pass, --write from a
`--pmc WRITE_SIZE` pass (each counter needs a pass of its own), --stats from `--kernel-trace --stats`.
Writes profiles/<tag>_traffic.json (HBM bytes per launch per kernel class, read by bench.py) and
profiles/<tag>_kernel_stats.csv (a copy of rocprofv3's kernel_stats.csv).

gfx950 correction: FETCH_SIZE counts 64 B for every 128-B request of the 16-byte-per-lane coalesced
reads these kernels issue, so read bytes = 2 x FETCH_SIZE (KB); WRITE_SIZE (KB) is taken as is.
"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_class(name: str):
    if "tree_step_kernel" in name: return "tree_step"
    if "tree_fused_kernel" in name: return "tree_fused"
    if "readout_kernel" in name: return "readout"
    if "conv_in_kernel" in name: return "conv0"
    if "conv_pair" in name or "conv_board" in name: return "conv2"       # conv0 + conv2 + conv4 in one launch; bench.py times it as conv2
    if "conv_rows" in name:
        return "conv2" if any(t in name for t in ("<7, 8", "<7,8", "ILi7ELi8E")) else "conv4"
    if "conv_mid" in name:
        return "conv2" if any(t in name for t in ("<7, 8", "<7,8", "ILi7ELi8E")) else "conv4"
    if "fc_f16_kernel" in name or "fc_kernel" in name or "fc_split_kernel" in name or "fc16_kernel" in name: return "fc"
    if "heads" in name: return "heads"
    return None


def counter_per_class(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = kernel_class(row["Kernel_Name"])
            if k:
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def all_counters_per_class(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for f in files:
        for row in csv.DictReader(open(f)):
            k = kernel_class(row["Kernel_Name"])
            if k:
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    return {k: {c: tot[k][c] / cnt[k][c] for c in tot[k]} for k in tot}, {k: max(cnt[k].values()) for k in cnt}


OUTDIR = os.path.join(ROOT, "profiles")


def write_l2_summary(d, tag):
    """Per-launch averages of a TCC_HIT_sum / TCC_MISS_sum pass and the L2 hit rate they give."""
    avg, launches = all_counters_per_class(d)
    path = os.path.join(OUTDIR, f"{tag}_pmc_l2_summary.csv")
    with open(path, "w") as fh:
        fh.write("kernel,launches,TCC_HIT_sum,TCC_MISS_sum,l2_hit_rate\n")
        for k in sorted(avg):
            h, m = avg[k].get("TCC_HIT_sum", 0.0), avg[k].get("TCC_MISS_sum", 0.0)
            fh.write(f"{k},{launches[k]},{h:.1f},{m:.1f},{h / (h + m) if h + m else 0.0:.4f}\n")
    print(open(path).read())


def write_sq_summary(d, tag):
    """Per-launch averages of an SQ/GRBM pass.  GRBM_GUI_ACTIVE is summed over the 8 XCDs, so
    cycles per launch = GRBM_GUI_ACTIVE / 8; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES (one count
    per SIMD-cycle an MFMA occupies) / (1024 SIMDs x cycles per launch)."""
    avg, launches = all_counters_per_class(d)
    names = sorted({c for k in avg for c in avg[k]})
    path = os.path.join(OUTDIR, f"{tag}_pmc_sq_summary.csv")
    with open(path, "w") as fh:
        fh.write("kernel,launches," + ",".join(names) + ",cycles_per_launch,mfma_util\n")
        for k in sorted(avg):
            cyc = avg[k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            util = avg[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else 0.0
            fh.write(f"{k},{launches[k]}," + ",".join(f"{avg[k].get(c, 0.0):.1f}" for c in names) + f",{cyc:.0f},{util:.4f}\n")
    print(open(path).read())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sq", help="directory of an SQ/GRBM counter pass (optional)")
    ap.add_argument("--l2", help="directory of a TCC_HIT_sum TCC_MISS_sum pass (optional)")
    ap.add_argument("--outdir", help="where the summaries go (default: profiles/)")
    ap.add_argument("--bench-log", help="stdout of the profiled bench.py run: its JSON line gives leaves per launch")
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--dtype", required=True)
    ap.add_argument("--tag", required=True)
    ap.add_argument("--command", default="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline")
    a = ap.parse_args()
    global OUTDIR
    if a.outdir:
        OUTDIR = a.outdir
    os.makedirs(OUTDIR, exist_ok=True)
    if a.sq:
        write_sq_summary(a.sq, a.tag)
    if a.l2:
        write_l2_summary(a.l2, a.tag)
    fetch, write = counter_per_class(a.fetch, "FETCH_SIZE"), counter_per_class(a.write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, n = fetch.get(k, (0.0, 0))
        w, _ = write.get(k, (0.0, 0))
        kernels[k] = {"launches": n, "fetch_size_kb": f, "write_size_kb": w, "hbm_bytes": 2.0 * f * 1024 + w * 1024}
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `{a.command}`",
           "correction": "gfx950: read bytes = 2 x FETCH_SIZE for 16-B-per-lane coalesced reads; WRITE_SIZE as reported",
           "unit": "bytes per launch", "dtype": a.dtype, "kernels": kernels}
    if a.bench_log:
        for line in open(a.bench_log):
            if line.startswith("{"):
                bj = json.loads(line)
                out["leaves_per_launch"] = bj["roofline"]["leaves_per_launch"]
                with open(os.path.join(OUTDIR, f"{a.tag}_bench_under_rocprof.json"), "w") as fh:
                    json.dump(bj, fh, indent=1)
    with open(os.path.join(OUTDIR, f"{a.tag}_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    if a.stats:
        st = glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(OUTDIR, f"{a.tag}_kernel_stats.csv"))
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
