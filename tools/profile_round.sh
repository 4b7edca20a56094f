#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + three PMC passes (HBM read, HBM write, matrix-pipe busy;
# one counter set per run, the program itself right after `--`) over the default training-step bench,
# then tools/profile_summary.py condenses them into gpurun_out/prof/summary_*.{csv,json} (copied to profiles/ by hand).
# usage: tools/profile_round.sh <tag> [extra bench.py flags, e.g. --workload gpt_vae_xl]
set -u
TAG=${1:-rXX}
shift || true
EXTRA="$*"
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras $EXTRA"
export PROFILE_CMD="$CMD"
# two kernel traces: the default step (a Block's weight gradients on a second stream: launches that share the chip last longer)
# and the single-stream step (MELGPT_WGRAD_SIDE=0: what bench.py's per_shape / frac_single_stream are measured on); the variable
# is exported, never passed through `env` (the program itself must follow `--`)
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace2" -o trace2 --output-format csv -- $CMD --no-reference-steps > "$OUT/trace2.log" 2>&1
export MELGPT_WGRAD_SIDE=0
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- $CMD > "$OUT/trace.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch --output-format csv -- $CMD > "$OUT/fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write --output-format csv -- $CMD > "$OUT/write.log" 2>&1
# matrix-pipe busy: SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA occupies a SIMD's matrix pipe, summed over SIMDs) against
# GRBM_GUI_ACTIVE (shader clock cycles of the dispatch, summed over the 8 XCDs) - SQ and GRBM slots are independent
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/mfma" -o mfma --output-format csv -- $CMD > "$OUT/mfma.log" 2>&1
# the in-kernel shader clock of the step's big kernels (diagnostic build with s_memtime / s_memrealtime stamps, built in the
# build container by tools/lab/build_clock_lib.py; MI355X_MICROARCH "DVFS give-back" item 6) -> summary "in_kernel_clock_GHz"
cd "$REPO"
if [ -f tools/lab/bin/libmelgpt_clock.so ] && [ -z "$EXTRA" ]; then
  MELGPT_LAB_LIB=$REPO/tools/lab/bin/libmelgpt_clock.so timeout 600 python3 tools/lab/clock_lab.py > "$OUT/clock_lab_$TAG.jsonl" 2> "$OUT/clock_lab.log"
fi
unset MELGPT_WGRAD_SIDE
PROFILE_STEPS=5 python3 tools/profile_summary.py "$OUT" "$TAG"
find "$OUT" -name "*.csv" -size +4M -delete   # the per-dispatch counter tables are too big to ship back
ls -la "$OUT"
