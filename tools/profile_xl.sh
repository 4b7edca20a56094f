set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_xl
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python3 $REPO/tools/bench_gpt_vae.py --steps 2 > "$OUT/trace.log" 2>&1
cd "$REPO"
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
find "$OUT" -name "*.csv" -size +4M -delete
tail -1 "$OUT/trace.log" | cut -c1-300
