#!/bin/bash
# Profiles one default bench.py round under rocprofv3: kernel trace + stats, then one pass per counter group.
# Usage (on the GPU box, from the repository root): bash tools/profile_round.sh <tag> [bench.py arguments]
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
# the raw traces are hundreds of megabytes: whatever happens, only the summaries and the logs travel back
trap 'rm -rf "$out/stats" "$out/fetch" "$out/write" "$out/sq" "$out/l2"; tail -n 5 "$out"/*.log' EXIT
args="--steps 1 --warmup 0 --no-cpu-baseline --no-extra-lines $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o run -- python3 bench.py $args > "$out/stats.log" 2>&1
timeout -s KILL 400 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$out/fetch" -o run -- python3 bench.py $args > "$out/fetch.log" 2>&1
timeout -s KILL 400 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$out/write" -o run -- python3 bench.py $args > "$out/write.log" 2>&1
timeout -s KILL 400 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d "$out/sq" -o run -- python3 bench.py $args > "$out/sq.log" 2>&1
timeout -s KILL 400 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$out/l2" -o run -- python3 bench.py $args > "$out/l2.log" 2>&1
prec=f32s
case " $* " in *" --precision f16 "*) prec=f16;; *" --precision f32 "*) prec=f32;; esac
python3 tools/summarize_pmc.py --fetch "$out/fetch" --write "$out/write" --stats "$out/stats" --sq "$out/sq" --l2 "$out/l2" \
    --dtype $prec --tag "$tag" --outdir "gpurun_out/profiles_$tag" --bench-log "$out/stats.log" \
    --command "python3 bench.py $args" > "$out/summary.log" 2>&1
echo profiled
