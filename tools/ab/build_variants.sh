#!/bin/bash
# Build libr0hip variants that differ only in how poseidon2.hip is compiled (A/B runs of hash_rows / hash_fold in ONE process on
# ONE box: tools/ab/ab_hash.py).  usage: tools/ab/build_variants.sh name1:"-Dflags" name2:"-Dflags" ...
set -e
cd "$(dirname "$0")/../.."
SRC=hyperfridge-r0_amd/csrc
OUT=tools/ab/lib
mkdir -p $OUT
make -s -j8 -C $SRC
OTHERS=$(ls $SRC/*.o | grep -v poseidon2.o)
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -ffp-contract=off $flags -c $SRC/poseidon2.hip -o $OUT/poseidon2_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libr0hip_$name.so $OTHERS $OUT/poseidon2_$name.o -L/opt/rocm/lib -lhiprtc -Wl,-rpath,/opt/rocm/lib
  echo built $OUT/libr0hip_$name.so "($flags)"
done
