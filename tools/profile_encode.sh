#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel trace of the VQ-encode alone -> gpurun_out/prof_enc/ (stats CSV)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_enc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python3 $REPO/tools/bench_encode.py --batch ${1:-128} --reps 5 > "$OUT/trace.log" 2>&1
cd "$REPO"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT" -name "*.csv" -size +4M -delete
tail -2 "$OUT/trace.log"
