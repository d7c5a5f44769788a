#!/bin/bash
# scratch/build_variant.sh NAME "EXTRA FLAGS" [translation units...]: scratch/libastk_NAME.so = the product objects with the named units
# (default: gemm) recompiled with EXTRA FLAGS -- for A/B runs inside one gpurun call (ASTK_LIB_PATH=scratch/libastk_NAME.so)
set -e
cd "$(dirname "$0")/../ast_amd/csrc"
NAME=$1; FL=$2; shift 2
UNITS=${@:-gemm}
mkdir -p ../_obj/var_$NAME
pids=()
for f in $UNITS; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $FL -c $f.hip -o ../_obj/var_$NAME/$f.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in util gemm rowgemm attn conv norm lstm lstm_persist decoder decoder_persist decoder_wide; do
  case " $UNITS " in *" $f "*) objs="$objs ../_obj/var_$NAME/$f.o";; *) objs="$objs ../_obj/$f.o";; esac
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libastk_$NAME.so $objs
echo built scratch/libastk_$NAME.so
