#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + two PMC passes over the default training-step bench,
# then tools/profile_summary.py condenses them into gpurun_out/prof/summary_*.{csv,json} (copied to profiles/ by hand).
# usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- $CMD > "$OUT/trace.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write --output-format csv -- $CMD > "$OUT/write.log" 2>&1
cd "$REPO" && python3 tools/profile_summary.py "$OUT" "$TAG"
find "$OUT" -name "*.csv" -size +4M -delete   # the per-dispatch counter tables are too big to ship back
ls -la "$OUT"
