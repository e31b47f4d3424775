#!/bin/bash
# an alternative libvfr build for same-lease A/B runs:  tools/build_variant.sh NAME "-DFLAG ..." file.hip [file.hip ...]
# -> video-fragments-retrieval_amd/lib/x_NAME.so (the named sources compiled with the extra flags, every other object as built
# by `make`); load it with VFR_LIB=.../x_NAME.so
set -e
name=$1; extra=$2; shift 2
cs=$(cd "$(dirname "$0")/../video-fragments-retrieval_amd/csrc" && pwd)
make -C "$cs" -j8 > /dev/null
tmp=$(mktemp -d)
objs=""
for o in "$cs"/*.o; do
  b=$(basename "$o" .o); use=$o
  for f in "$@"; do
    if [ "$(basename "$f" .hip)" = "$b" ]; then
      fl="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wno-unused-function"
      [ "$b" = score ] && fl="$fl -fno-slp-vectorize"
      /opt/rocm/bin/hipcc $extra $fl -I"$cs" -c "$cs/$b.hip" -o "$tmp/$b.o"; use=$tmp/$b.o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$cs/../lib/x_$name.so" $objs
rm -rf "$tmp"; echo "built $cs/../lib/x_$name.so"
