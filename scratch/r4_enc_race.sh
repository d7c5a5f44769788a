#!/bin/bash
# which variant of the encoder backward shows the rare stale hand-off (scratch/enc_repeat.py with alternating inputs)
cd "$GRAFT_REPO_ROOT"
for lib in ${LIBS:-ast_amd/libastk.so scratch/vob/libastk_FENCE.so}; do
  echo "== $lib"
  ASTK_LIB_PATH=$PWD/$lib timeout -k 10 400 python3 scratch/enc_repeat.py ${N:-12000} 2>&1 | tail -n 2 | cut -c1-300
done
