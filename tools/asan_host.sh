#!/bin/bash
# Host-side AddressSanitizer build of the C-ABI launch layer (SURVEY 5 / the review's "sanitizer build" item): every
# csrc/*.hip compiled with -fsanitize=address for the HOST code only (-fno-gpu-sanitize: GPU ASan is not available on this
# pool), linked into a scratch library OUTSIDE the tree (the product library is never an ASan build), and driven by
# tools/asan_host_driver.c.  Needs no GPU.  usage: tools/asan_host.sh [jobs]   (a few minutes: -O1, device code included)
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${MELGPT_ASAN_DIR:-/tmp/melgpt_asan}
JOBS=${1:-6}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$OUT"
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-inline-asm -Wno-unused-value -ffp-contract=fast -fno-gpu-rdc -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -I $REPO/include"
ls "$REPO"/melspec_gpt_vqvae_amd/csrc/*.hip | xargs -P "$JOBS" -I{} sh -c "$HIPCC $FLAGS -c {} -o $OUT/\$(basename {} .hip).o"
$HIPCC -shared -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -o "$OUT/libmelgpt_hip_asan.so" "$OUT"/*.o
$HIPCC -x c -O1 -g -fsanitize=address -fno-omit-frame-pointer -I "$REPO/include" "$REPO/tools/asan_host_driver.c" -L "$OUT" -lmelgpt_hip_asan -Wl,-rpath,"$OUT" -o "$OUT/asan_host_driver"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 "$OUT/asan_host_driver"
