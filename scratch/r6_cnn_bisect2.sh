#!/bin/bash
cd $GRAFT_REPO_ROOT
T='tests/test_gpu_ops.py::test_cnn_fwd_bwd[fp16x2-False-16-400-13-128-512]'
run() { echo "== $1"; timeout -k 10 120 python -c "
import torch, sys
from ast_amd import _lib
for kv in '$1'.split(','):
    if kv: _lib.set_tuning(kv.split('=')[0], float(kv.split('=')[1]))
import pytest
sys.exit(pytest.main(['-q', '-x', '$T']))" 2>&1 | grep -E "passed|failed|max abs err|astk_gemm" | head -${2:-3}; }
run "gemm.forward_pairs=0"
run "gemm.log=1" 60
