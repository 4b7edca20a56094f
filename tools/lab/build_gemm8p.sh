#!/bin/bash
# LAB: build tools/lab/gemm8p_lab.hip into tools/lab/bin/libgemm8p.so (extra flags, e.g. -DP8_PRIO=0, are passed through)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$ROOT/tools/lab/bin"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wno-inline-asm "$@" \
  -I "$ROOT/melspec_gpt_vqvae_amd/csrc" "$ROOT/tools/lab/gemm8p_lab.hip" -o "${OUT:-$ROOT/tools/lab/bin/libgemm8p.so}"
