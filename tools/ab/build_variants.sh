#!/bin/bash
# Build libr0hip variants that differ only in how ONE translation unit is compiled -- poseidon2.hip by default (A/B runs of hash_rows /
# hash_fold in ONE process on ONE box: tools/ab/ab_hash.py), UNIT=ntt for ntt.hip (tools/ab/ab_ntt.py).
# usage: [UNIT=ntt] tools/ab/build_variants.sh name1:"-Dflags" name2:"-Dflags" ...
set -e
cd "$(dirname "$0")/../.."
SRC=hyperfridge-r0_amd/csrc
OUT=tools/ab/lib
mkdir -p $OUT
make -s -j8 -C $SRC
UNIT=${UNIT:-poseidon2}
OTHERS=$(ls $SRC/*.o | grep -v "/$UNIT.o")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -ffp-contract=off $flags -c $SRC/$UNIT.hip -o $OUT/${UNIT}_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libr0hip_$name.so $OTHERS $OUT/${UNIT}_$name.o -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib
  echo built $OUT/libr0hip_$name.so "($flags)"
done
