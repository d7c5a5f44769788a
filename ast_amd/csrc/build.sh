#!/bin/bash
# Builds (gfx950 only, in-tree):
#   ast_amd/libastk.so       the product library
#   ast_amd/libastk_test.so  the same sources with -DASTK_TEST_HOOKS: adds the test instrumentation entry points (astk_conv_debug_*,
#                            astk_debug_set_amax_generation) that the product library does not export; loaded by tests only
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $ASTK_EXTRA_FLAGS"
SRCS="util gemm rowgemm attn conv norm lstm lstm_persist decoder decoder_persist decoder_wide"
HOOKED="gemm conv lstm_persist decoder_persist"        # the translation units with #ifdef ASTK_TEST_HOOKS sections
mkdir -p ../_obj ../_obj/test
pids=()
stale() { [ ! -f $2 ] || [ $1.hip -nt $2 ] || [ common.h -nt $2 ] || [ decoder_wide.h -nt $2 ] || [ ../../include/astk.h -nt $2 ]; }
for f in $SRCS; do
  if stale $f ../_obj/$f.o; then hipcc $FLAGS -c $f.hip -o ../_obj/$f.o & pids+=($!); fi
done
for f in $HOOKED; do
  if stale $f ../_obj/test/$f.o; then hipcc $FLAGS -DASTK_TEST_HOOKS -c $f.hip -o ../_obj/test/$f.o & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p; done
objs=""; tobjs=""
for f in $SRCS; do
  objs="$objs ../_obj/$f.o"
  case " $HOOKED " in *" $f "*) tobjs="$tobjs ../_obj/test/$f.o";; *) tobjs="$tobjs ../_obj/$f.o";; esac
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libastk.so $objs
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libastk_test.so $tobjs
echo "built $(realpath ../libastk.so) and $(realpath ../libastk_test.so)"
