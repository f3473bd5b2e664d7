#!/usr/bin/env bash
# build_variant.sh NAME [hipcc flags]: the library with extra flags as tools/prof/libs/libdskd_NAME.so (for DSKD_HIP_LIB A/B runs)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
name=$1; shift
tmp=$(mktemp -d)
src="$here/../../dskd_amd/csrc"
mkdir -p "$here/libs"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -fno-fast-math -ffp-contract=on -Wno-unused-function"
objs=()
for f in "$src"/*.hip "$src"/capi.cpp; do
  b=$(basename "$f"); o="$tmp/${b%.*}.o"
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$f" -o "$o" &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$here/libs/libdskd_$name.so"
rm -rf "$tmp"
echo "built tools/prof/libs/libdskd_$name.so"
