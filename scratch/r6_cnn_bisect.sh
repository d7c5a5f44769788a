#!/bin/bash
cd $GRAFT_REPO_ROOT
T='tests/test_gpu_ops.py::test_cnn_fwd_bwd[fp16x2-False-16-400-13-128-512]'
run() { echo "== $1 lib=${2:-current}"; ASTK_LIB_PATH=$2 timeout -k 10 120 python -c "
import torch, sys
from ast_amd import _lib
for kv in '$1'.split(','):
    if kv: _lib.set_tuning(kv.split('=')[0], float(kv.split('=')[1]))
import pytest
sys.exit(pytest.main(['-q', '-x', '$T']))" 2>&1 | grep -E "passed|failed|max abs err" | head -3; }
run ""
run "gemm.chunk=0"
run "gemm.hybrid=0"
run "colreduce.blocks=64"
run "conv.seq_bwd=0"
run "conv.seq_fwd=0"
for c in e9ad643 1fdd57b 4faccd3; do [ -f scratch/prev_tree/$c/ast_amd/libastk.so ] && run "" $GRAFT_REPO_ROOT/scratch/prev_tree/$c/ast_amd/libastk.so; done
