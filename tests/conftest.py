import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def tiny_cfg(enc_layers=2, dec_layers=2, H=8, E=4, A=8, c0=4, c1=6, V=11, drop=0.0):
    """A miniature of experiments/es_en_20h/model_cfg.json (same structure, small widths)."""
    return {"dropout": {"embed": drop, "rnn": drop, "out": 0},
            "rnn_config": {"bi_rnn": True, "enc_layers": enc_layers, "dec_layers": dec_layers, "hidden_units": H,
                           "embedding_units": E, "attn_units": A, "n_attn": 1, "feed_attn": True, "ln": False,
                           "dec_vocab_size": V},
            "cnn_config": {"bn": True, "cnn_layers": [
                {"in_channels": None, "out_channels": c0, "ksize": [9, 13], "stride": [2, 13], "pad": [4, 0]},
                {"in_channels": None, "out_channels": c1, "ksize": [9, 1], "stride": [2, 1], "pad": [4, 0]}]}}


@pytest.fixture
def tiny():
    return tiny_cfg


SCHEMES = ["bf16x3", "f32", "fp16x2"]      # include/astk.h ASTK_PREC_*: the library default first, then the literal f32 chain, then the opt-in narrow scheme


@pytest.fixture(params=SCHEMES)
def gemm_scheme(request):
    """The whole-model parity tests run once under EVERY arithmetic scheme bench.py times (round-3 review, item 1b): the test sets
    `model.gemm_precision = gemm_scheme`, which goes into the op descriptors' `precision` field (per call, no process-wide state)."""
    return request.param


@pytest.fixture
def tune():
    """tune(key, value, lib=None): sets one of the library's documented tuning knobs (astk_set_tuning, include/astk.h) for the running
    test -- on `lib` or on the currently loaded library -- and puts the old value back afterwards.  (Rounds 1-5 used environment
    variables for this; the product library reads none any more: tests/test_host.py checks.)"""
    import ctypes as C
    from ast_amd import _lib
    saved = []

    def _set(key, value, lib=None):
        lib = lib or _lib.load()
        prev = C.c_double()
        assert lib.astk_get_tuning(key.encode(), C.byref(prev)) == 0, key
        assert lib.astk_set_tuning(key.encode(), float(value)) == 0, key
        saved.append((lib, key, prev.value))
    yield _set
    for lib, key, value in reversed(saved):
        lib.astk_set_tuning(key.encode(), value)
