#!/usr/bin/env bash
# Builds the gfx950 hot-path library in-tree: dskd_amd/_C/libdskd_hip.so
# (hipcc cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../_C"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -fno-fast-math -ffp-contract=on -Wall -Wno-unused-function"
objs=()
for f in msda msda_mm msda_pull msda_prep denseloss addln biasact ffnact ffn_mfma gemm_nt optim winattn attn gn lsap cost corr fgkd; do
  "$HIPCC" $FLAGS "$@" -c "$here/$f.hip" -o "$out/$f.o" &
  objs+=("$out/$f.o")
done
"$HIPCC" $FLAGS "$@" -c "$here/capi.cpp" -o "$out/capi.o" &
objs+=("$out/capi.o")
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/libdskd_hip.so"
echo "built $out/libdskd_hip.so"
