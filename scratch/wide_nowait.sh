#!/bin/bash
# timing experiment: the wide decoder kernels with their waits compiled out (results are wrong): work per step without hand-offs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in ast_amd/libastk.so scratch/libastk_nowait.so; do
  O=gpurun_out/ks_nw; rm -rf $O; mkdir -p $O
  ASTK_LIB_PATH=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --model cfg5 --steps 6 --warmup 2 --no-cpu-baseline --no-alt-precisions --profile-steps 0 > $O/bench.log 2>&1
  echo $lib; python3 scratch/kstat2.py $O 8 60 | grep "decoder_wide"
done
