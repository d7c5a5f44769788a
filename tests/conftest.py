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
