#!/bin/bash
# Like build_variant.sh, but recompiles only the named sources with the extra flags and links them with the
# objects of the main build (make -C spmf_amd/csrc first).
# usage: tools/build_variant_fast.sh <name> "<extra flags>" file1 [file2 ...]     (files without .hip)
set -e
name=$1; extra=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/spmf_amd/csrc
out=$root/spmf_amd/variants
tmp=$(mktemp -d)
mkdir -p $out
pids=()
for f in "$@"; do
  per=""
  [ $f = dense3 ] && per="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -DSPMF_BUILD --offload-arch=gfx950 \
      -Wno-unused-function -I$root/include -I$src $per $extra -c $src/$f.hip -o $tmp/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build_variant_fast: a compile failed"; rm -rf $tmp; exit 1; }; done
objs=""
for o in $src/*.o; do
  b=$(basename $o .o)
  case "$b" in *_asan) continue;; esac
  if [ -f $tmp/$b.o ]; then objs="$objs $tmp/$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=$src/exports.map -Wl,--no-undefined $objs -o $out/libspmf_$name.so -ldl
rm -rf $tmp
echo built $out/libspmf_$name.so
