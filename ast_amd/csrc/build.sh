#!/bin/bash
# Builds libastk.so (gfx950 only) in-tree: ast_amd/libastk.so
set -e
cd "$(dirname "$0")"
OUT=../libastk.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $ASTK_EXTRA_FLAGS"
mkdir -p ../_obj
pids=()
for f in util gemm rowgemm attn conv norm lstm lstm_persist decoder decoder_persist decoder_wide; do
  if [ ! -f ../_obj/$f.o ] || [ $f.hip -nt ../_obj/$f.o ] || [ common.h -nt ../_obj/$f.o ] || [ decoder_wide.h -nt ../_obj/$f.o ] || [ ../../include/astk.h -nt ../_obj/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o ../_obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../_obj/util.o ../_obj/gemm.o ../_obj/rowgemm.o ../_obj/attn.o ../_obj/conv.o ../_obj/norm.o ../_obj/lstm.o ../_obj/lstm_persist.o ../_obj/decoder.o ../_obj/decoder_persist.o ../_obj/decoder_wide.o
echo "built $(realpath $OUT)"
