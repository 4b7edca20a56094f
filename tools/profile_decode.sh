#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations of the KV-cached decode step at batch $1 (default 64) -> gpurun_out/prof_dec/
set -u
B=${1:-64}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_dec
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/trace$B" -o trace --output-format rocpd -- python3 $REPO/tools/profile_decode.py $B > "$OUT/trace$B.log" 2>&1
cd "$REPO"
DB=$(find "$OUT/trace$B" -name "*.db" | head -1)
python3 tools/decode_trace_stats.py "$DB" ${2:-124} > "$OUT/decode_B$B.txt" 2>&1
grep ms_per_token "$OUT/trace$B.log"
cat "$OUT/decode_B$B.txt"
find "$OUT" -name "*.db" -size +30M -delete
