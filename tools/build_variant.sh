#!/bin/bash
# Build libspmf_hip with extra -D flags into spmf_amd/variants/libspmf_<name>.so
# (kernel experiments: select with SPMF_LIB_PATH; *.so is git-ignored but travels with gpurun).
# usage: tools/build_variant.sh <name> "<extra flags>"
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/spmf_amd/csrc
out=$root/spmf_amd/variants
tmp=$(mktemp -d)
mkdir -p $out
pids=()
for f in api prep row_pass col_pass finish stats dense dense3 dense_ll surrogate layout p2p widek; do
  per=""
  [ $f = dense3 ] && per="-fno-slp-vectorize"      # as in csrc/Makefile
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -DSPMF_BUILD --offload-arch=gfx950 \
      -Wno-unused-function -I$root/include -I$src $per $extra -c $src/$f.hip -o $tmp/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "build_variant: a compile failed"; rm -rf $tmp; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=$src/exports.map -Wl,--no-undefined $tmp/*.o -o $out/libspmf_$name.so -ldl
rm -rf $tmp
echo built $out/libspmf_$name.so
