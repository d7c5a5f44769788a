"""CPU tests of the host-side mirror of the reference interface: config, bucketing, batch contract, parameter
layout, checkpoints' shape inference, and that libastk.so loads and exports every symbol include/astk.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import json
import os
import pickle
import random
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIPPED = {"dropout": {"embed": 0.3, "rnn": 0.3, "out": 0},
           "rnn_config": {"bi_rnn": True, "enc_layers": 3, "dec_layers": 3, "hidden_units": 512, "embedding_units": 128,
                          "attn_units": 512, "n_attn": 1, "feed_attn": True, "ln": False},
           "cnn_config": {"bn": True, "cnn_layers": [
               {"in_channels": None, "out_channels": 128, "ksize": [9, 13], "stride": [2, 13], "pad": [4, 0]},
               {"in_channels": None, "out_channels": 512, "ksize": [9, 1], "stride": [2, 1], "pad": [4, 0]}]}}


def test_library_exports_every_declared_symbol():
    """include/astk.h is the boundary: the product library exports exactly the symbols it declares outside its #ifdef ASTK_TEST_HOOKS
    sections; the test instrumentation declared inside them (astk_conv_debug_*, astk_debug_*) exists in libastk_test.so only."""
    from ast_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "astk.h")).read()
    hook_sections = re.findall(r"#ifdef ASTK_TEST_HOOKS(.*?)#endif", hdr, flags=re.S)
    product_hdr = re.sub(r"#ifdef ASTK_TEST_HOOKS.*?#endif", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(astk_[a-z0-9_]+)\s*\(", product_hdr))
    hooks = set(re.findall(r"\b(astk_[a-z0-9_]+)\s*\(", "\n".join(hook_sections)))
    assert len(declared) >= 25
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert hooks == set(_lib.TEST_HOOK_SIGNATURES) and hooks, hooks ^ set(_lib.TEST_HOOK_SIGNATURES)
    lib = _lib.load()                                   # raises if the .so or a symbol is missing
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)
    for name in hooks:                                  # debug hooks are not part of the product ABI
        assert not hasattr(lib, name), f"libastk.so exports the test hook {name}"
    assert "astk_conv_debug_kill_units" in hooks
    with _lib.load_test_hooks() as tlib:                # ... and the instrumented build has everything
        for name in declared | hooks:
            assert isinstance(getattr(tlib, name), ctypes._CFuncPtr)
        assert _lib.load() is tlib
    assert _lib.load() is lib
    assert lib.astk_version() >= 100
    # in-kernel instrumentation of the persistent kernels (phase timers, the dawdling slice of the last-arrival regression test) is read
    # from ASTK_PERSIST_DBG by the TEST-HOOK build only: the product library does not even contain the variable's name
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"ASTK_PERSIST_DBG" not in blob
    assert b"ASTK_PERSIST_DBG" in open(_lib.TEST_LIB_PATH, "rb").read()
    # ... and the product library reads NO environment variable at all (round 6: the ~30 getenv knobs of rounds 1-5 are entries of the one
    # documented switchboard astk_set_tuning): no ASTK_ name of that kind is in the binary, and it does not import getenv
    import subprocess
    names = set(re.findall(rb"ASTK_[A-Z][A-Z0-9_]{2,}", blob))
    assert not names, names
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        assert not re.search(r"\bU (secure_)?getenv\b", nm.stdout), "libastk.so imports getenv"
    # every knob is enumerable and round-trips
    import ctypes as C
    keys = []
    while lib.astk_tuning_key(len(keys)):
        keys.append(lib.astk_tuning_key(len(keys)).decode())
    assert {"gemm.hybrid", "dec.persist", "lstm.persist", "persist.spin_limit", "lstm.rows32"} <= set(keys) and len(keys) == len(set(keys))
    for k in keys:
        v = C.c_double()
        assert lib.astk_get_tuning(k.encode(), C.byref(v)) == 0
        assert _lib.set_tuning(k, v.value) == v.value
    assert lib.astk_set_tuning(b"no.such.knob", 1.0) != 0 and b"no.such.knob" in lib.astk_last_error()
    # ... every knob is DOCUMENTED in the header, and setting one moves no other (the table's rows carry their key: util.hip's static_assert
    # keeps a name from landing on another knob's slot, which no outside test can see -- set and get would agree on the wrong slot)
    hdr = open(os.path.join(ROOT, "include", "astk.h")).read()
    assert not [k for k in keys if k not in hdr], [k for k in keys if k not in hdr]
    def read_all():
        out = {}
        for q in keys:
            v = C.c_double()
            assert lib.astk_get_tuning(q.encode(), C.byref(v)) == 0
            out[q] = v.value
        return out
    before = read_all()
    for k in ("lstm.side_bwd", "lstm.duo_side", "gemm.forward_pairs"):
        prev = _lib.set_tuning(k, 7.0)
        assert read_all() == {**before, k: 7.0}, k
        _lib.set_tuning(k, prev)
    assert read_all() == before


def test_parameter_counts_match_the_reference_model():
    from ast_amd.params import param_shapes
    for D, n in ((13, 12333258), (80, 17576138)):       # SURVEY.md 8(a) row a3 / BASELINE.md
        train, persist = param_shapes(SHIPPED, D, 1098)
        assert sum(int(np.prod(s)) for s in train.values()) == n
        assert set(persist) == {f"CNN_{i}_bn/{s}" for i in (0, 1) for s in ("avg_mean", "avg_var")}
    train, _ = param_shapes(SHIPPED, 80, 1098)
    assert train["L0_enc/upward/W"] == (1024, 3072) and train["L0_dec/upward/W"] == (2048, 640)
    assert train["context/W"] == (512, 1024) and train["embed_dec/W"] == (1098, 128)


def test_initialisers_follow_A9():
    from ast_amd.params import init_values
    cfg = json.loads(json.dumps(SHIPPED))
    cfg["rnn_config"].update(hidden_units=64, attn_units=32, embedding_units=16)
    v = init_values(cfg, 13, 50, seed=1)
    b = v["L1_rev_enc/upward/b"]
    assert (b[2::4] == 1).all() and b.sum() == b[2::4].sum()
    assert abs(v["CNN_1/W"].std() - np.sqrt(2.0 / (128 * 9))) < 0.1 * np.sqrt(2.0 / (128 * 9))
    assert abs(v["embed_dec/W"].std() - 1.0) < 0.1
    assert (v["CNN_0_bn/gamma"] == 1).all() and (v["CNN_0_bn/avg_var"] == 1).all() and (v["CNN_0_bn/avg_mean"] == 0).all()


def test_arena_layout_is_16_byte_aligned_and_flat():
    import torch
    from ast_amd.params import ParamArena
    a = ParamArena({"a/W": (3, 5), "a/b": (7,), "c/W": (2, 2)}, torch.device("cpu"))
    assert [a.offsets[k] for k in ("a/W", "a/b", "c/W")] == [0, 16, 24] and a.size == 28
    a.views["a/b"].fill_(2.0)
    assert float(a.data.sum()) == 14.0 and a.data[16:23].eq(2).all() and a.data[23] == 0
    assert a.range_of("a/W") == (0, 16)


GOLD = os.path.join(ROOT, "tests", "golden")


def test_bucket_histogram_of_the_reference_corpus_description():
    """preprocessing/prep_buckets.py:41-63 on the reference-held data/fisher/fisher_20h.info (frame counts extracted into
    tests/golden/fisher_20h_frames.json by make_fisher_frames.py): the Fisher-20h train histogram SURVEY.md 8(a) row a1 quotes, the
    corpus totals of SURVEY.md section 6, and the bucket the benchmark's "T ~ 800" comes from."""
    from ast_amd import prep_buckets
    d = json.load(open(os.path.join(GOLD, "fisher_20h_frames.json")))
    frames = d["frames"]["fisher_train"]
    assert len(frames) == 17306 and sum(frames) == 7238296 and max(frames) == 2566
    info = {"fisher_train": {f"u{i}": {"sp": t} for i, t in enumerate(frames)},
            "fisher_dev": {f"d{i}": {"sp": t} for i, t in enumerate(d["frames"]["fisher_dev"])}}
    out = prep_buckets.buckets_from_info(info, 20, 80, "sp", scale=1, seed="seed-ast-20h")
    hist = [len(b) for b in out["fisher_train"]["buckets"]]
    assert hist == [1025, 3516, 2543, 1939, 1486, 1188, 932, 736, 674, 603, 550, 505, 420, 342, 277, 189, 138, 86, 63, 94]
    assert sum(len(b) for b in out["fisher_dev"]["buckets"]) == 3977
    # T ~ 800: buckets 9 and 10 hold the 720..879-frame utterances; everything above 1680 frames is truncated by the loader
    assert all(720 <= info["fisher_train"][u]["sp"] < 800 for u in out["fisher_train"]["buckets"][9])
    assert sum(t > 1680 for t in frames) == sum(1 for t in frames if t // 80 >= 21)
    assert d["vocab_types"]["bpe_w"] == 1098                   # dec_vocab_size of the shipped experiment (config.py:24)


@pytest.mark.parametrize("exp,V,D", [("es_en_20h", 1098, 13), ("asr_gpfr", 1004, 13)])
def test_reference_experiment_configs_load_through_config_and_model(tmp_path, exp, V, D):
    """The reference's ACTUAL experiments/<exp>/{model,train}_cfg.json (committed as data fixtures under
    tests/golden/ref_experiments/) go through Config (config.py:15-29), the model constructor (seq2seq.py:23-156: every key it reads)
    and the optimizer set-up (nn.py:81-119); asr_gpfr has no n_attn / feed_attn keys (defaults) and dataloader = globalphone."""
    import shutil
    from ast_amd.config import Config
    from ast_amd.params import param_shapes
    from ast_amd.seq2seq import SpeechEncoderDecoder
    from ast_amd import optimizers as O
    for f in ("model_cfg.json", "train_cfg.json"):
        shutil.copy(os.path.join(GOLD, "ref_experiments", exp, f), tmp_path / f)
    tcfg = json.load(open(tmp_path / "train_cfg.json"))
    dec_key = tcfg["data"]["dec_key"]
    vocab = {dec_key: {"w2i": {str(i).encode(): i for i in range(V)}, "i2w": {}, "freq": {}}}
    pickle.dump(vocab, open(tmp_path / "v.vocab", "wb"))
    tcfg["data"]["vocab_path"] = str(tmp_path / "v.vocab")     # the vocab pickle itself is not in the reference tree's experiments
    json.dump(tcfg, open(tmp_path / "train_cfg.json", "w"))
    c = Config(str(tmp_path))
    assert c.model["rnn_config"]["dec_vocab_size"] == V and c.model["model_dir"] == str(tmp_path)
    assert c.train["seed"] == "seed-ast-20h" and c.train["batch_size"] == 32 and c.train["extras"]["teach_ratio"] == 0.8
    assert c.train["data"]["buckets_num"] == 20 and c.train["data"]["buckets_width"] == 80 and c.train["data"]["max_pred"] == 175
    m = SpeechEncoderDecoder(-1, c.model)                      # constructor only: no GPU here
    assert (len(m.rnn_enc), len(m.rnn_rev_enc), len(m.rnn_dec), m.H, m.h, m.E, m.A) == (3, 3, 3, 512, 256, 128, 512)
    assert m.rnn_dec == ["L0_dec", "L1_dec", "L2_dec"] and m.cnns == ["CNN_0", "CNN_1"]
    train, persist = param_shapes(c.model, D, V)
    assert train["L0_enc/upward/W"] == (1024, 512) and train["L2_dec/upward/W"] == (2048, 512) and train["out/W"] == (V, 512)
    if exp == "es_en_20h":
        assert sum(int(np.prod(s)) for s in train.values()) == 12333258
    oc = c.train["optimizer"]
    assert (oc["type"], oc["lr"], oc["l2"], oc["grad_clip"], oc["grad_noise_eta"], oc["freeze"]) == (0, 0.001, 0.0001, 2, 0, [])
    opt = O.Adam(alpha=oc["lr"], beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True)
    opt.add_hook(O.WeightDecay(oc["l2"]))
    opt.add_hook(O.GradientClipping(oc["grad_clip"]))
    assert [type(h).__name__ for h in opt.hooks] == ["WeightDecay", "GradientClipping"]


def test_bucketing_matches_prep_buckets():
    from ast_amd import prep_buckets
    info = {"fisher_train": {f"u{i}": {"sp": t} for i, t in enumerate([27, 79, 80, 159, 160, 1599, 1600, 2566])},
            "fisher_dev": {"d0": {"sp": 300}}}
    out = prep_buckets.buckets_from_info(info, 20, 80)
    b = out["fisher_train"]["buckets"]
    assert b[0] == ["u0", "u1"] and b[1] == ["u2", "u3"] and b[2] == ["u4"] and b[19] == ["u5", "u6", "u7"]
    assert out["fisher_dev"]["buckets"][3] == ["d0"] and out["fisher_train"]["width_b"] == 80
    # train_scale > 1 subsamples train sets only, with the fixed seed
    big = {"x_train": {f"u{i}": {"sp": 10} for i in range(10)}, "dev": {f"d{i}": {"sp": 10} for i in range(10)}}
    o2 = prep_buckets.buckets_from_info(big, 2, 80, scale=2)
    assert len(o2["x_train"]["buckets"][0]) == 5 and len(o2["dev"]["buckets"][0]) == 10
    assert o2 == prep_buckets.buckets_from_info(big, 2, 80, scale=2)


def _synth_cfg(tmp_path, n_train=37):
    data = {"dataloader": "synthetic", "vocab_size": 31, "feat_dim": 13, "n_utts": {"syn_train": n_train, "syn_dev": 5},
            "frames": [20, 400], "targets": [1, 30], "buckets_num": 4, "buckets_width": 80, "max_pred": 12,
            "zero_input": 0.1, "train_scale": 1, "dec_key": "bpe_w"}
    return data


def test_batch_contract(tmp_path):
    from ast_amd.dataloader import SYMBOLS, SyntheticDataLoader
    data = _synth_cfg(tmp_path)
    dl = SyntheticDataLoader(data, str(tmp_path), -1)
    assert dl.n_utts == {"syn_train": 37, "syn_dev": 5}
    random.seed("seed-ast-20h")
    plan1 = dl.batch_plan(8, "syn_train")
    random.seed("seed-ast-20h")
    dl2 = SyntheticDataLoader(data, str(tmp_path), -1)
    assert [u for u, _ in dl2.batch_plan(8, "syn_train")] == [u for u, _ in plan1]      # seeded stream => same plan
    random.seed(1)
    seen = []
    for batch in dl.get_batch(8, "syn_train", train=True, labels=True):
        X, y = batch["X"], batch["y"]
        assert X.dtype.is_floating_point and X.shape[2] == 13 and y.dtype == __import__("torch").int32
        assert X.shape[0] == y.shape[0] == len(batch["utts"]) <= 8
        assert X.shape[1] <= (4 + 1) * 80                          # hard truncation (dataloader.py:118)
        widths = {min(dl.info["syn_train"][u]["sp"] // 80, 3) for u in batch["utts"]}
        assert len(widths) == 1                                    # one bucket per batch
        yn = y.numpy()
        assert (yn[:, 0] == SYMBOLS.GO_ID).all() and y.shape[1] <= 12
        for row in yn:
            n = int((row != 0).sum())
            assert row[n - 1] == SYMBOLS.EOS_ID and (row[n:] == SYMBOLS.PAD_ID).all()
        for i, u in enumerate(batch["utts"]):                      # zero padding beyond each utterance
            t = min(dl.info["syn_train"][u]["sp"], 400)
            assert float(X[i, t:].abs().sum()) == 0.0
            zero_rows = int((X[i, :t].abs().sum(dim=1) == 0).sum())
            assert zero_rows <= int(0.1 * t)                       # frame zeroing, with replacement
        seen += batch["utts"]
    assert sorted(seen) == sorted(dl.info["syn_train"])
    dev = next(dl.get_batch(8, "syn_dev", train=False, labels=False))
    assert "y" not in dev


def test_data_parallel_sharding_of_batches(tmp_path):
    """Rank r takes rows r::world of every bucketed batch, padded to the extents of the WHOLE batch: the shards are disjoint, equally
    large, and have the same (T, L) on every rank -- equal work and, because one teacher-forcing coin is drawn per decoder step, the same
    consumption of the seeded `random` stream (otherwise the ranks' shuffles would drift apart after the first batch)."""
    from ast_amd.dataloader import SyntheticDataLoader
    data = _synth_cfg(tmp_path, n_train=64)
    shards = []
    for r in range(2):
        dl = SyntheticDataLoader(data, str(tmp_path), -1)
        dl.rank, dl.world = r, 2
        random.seed("seed-ast-20h")
        shards.append([(b["utts"], tuple(b["X"].shape[1:]), b["y"].shape[1]) for b in dl.get_batch(8, "syn_train", train=True, labels=True)])
    flat0, flat1 = sum((u for u, _, _ in shards[0]), []), sum((u for u, _, _ in shards[1]), [])
    # disjoint; a bucket's last batch is cut to a multiple of the world size, so at most one utterance per bucket sits out (4 buckets)
    assert not set(flat0) & set(flat1) and 64 - 4 <= len(flat0) + len(flat1) <= 64
    # the same number of steps, the same shard size and the same extents on every rank, step by step
    assert len(shards[0]) == len(shards[1])
    assert [(len(u), x, l) for u, x, l in shards[0]] == [(len(u), x, l) for u, x, l in shards[1]]
    # evaluation is not sharded: the rank that scores the dev set sees all of it
    assert sorted(sum((b["utts"] for b in dl.get_batch(8, "syn_dev", train=False, labels=False)), [])) == sorted(dl.info["syn_dev"])


def test_evaluation_on_rank0_only_keeps_the_ranks_plans_in_step(tmp_path):
    """train.py under data parallelism: only rank 0 runs the dev pass between two epochs.  Its batch plan consumes the seeded
    `random` stream (and shuffles bucket lists in place); the pass must leave both as it found them, or rank 0's shards,
    pads and teacher-forcing flags differ from the other ranks' from the second epoch on."""
    from ast_amd.dataloader import SyntheticDataLoader
    data = _synth_cfg(tmp_path, n_train=64)
    plans = []
    for r in range(2):
        dl = SyntheticDataLoader(data, str(tmp_path), -1)
        dl.rank, dl.world = r, 2
        random.seed("seed-ast-20h")
        epoch1 = [b["utts"] for b in dl.get_batch(8, "syn_train", train=True, labels=True)]
        if r == 0:                                    # the dev pass, and an evaluation over the TRAIN set for good measure
            list(dl.get_batch(8, "syn_dev", train=False, labels=False))
            list(dl.get_batch(8, "syn_train", train=False, labels=False))
        coin = random.random()                        # where the stream stands: the next teacher-forcing coin
        epoch2 = [(b["utts"], tuple(b["X"].shape[1:]), b["y"].shape[1]) for b in dl.get_batch(8, "syn_train", train=True, labels=True)]
        plans.append((epoch1, coin, epoch2))
    assert plans[0][1] == plans[1][1]
    assert [(len(u), x, l) for u, x, l in plans[0][2]] == [(len(u), x, l) for u, x, l in plans[1][2]]
    assert not set(sum((u for u, _, _ in plans[0][2]), [])) & set(sum((u for u, _, _ in plans[1][2]), []))


def test_config_injects_vocab_size(tmp_path):
    from ast_amd.config import Config
    vocab = {"bpe_w": {"w2i": {bytes([i]): i for i in range(57)}, "i2w": {}, "freq": {}}}
    vp = tmp_path / "v.vocab"
    pickle.dump(vocab, open(vp, "wb"))
    json.dump(SHIPPED, open(tmp_path / "model_cfg.json", "w"))
    json.dump({"seed": "s", "gpuid": 0, "data": {"vocab_path": str(vp), "dec_key": "bpe_w"}}, open(tmp_path / "train_cfg.json", "w"))
    c = Config(str(tmp_path))
    assert c.model["rnn_config"]["dec_vocab_size"] == 57 and c.model["model_dir"] == str(tmp_path)


def test_checkpoint_shape_inference():
    from ast_amd.serializers import infer_in_dim
    assert infer_in_dim(SHIPPED, 3072) == 80 - 2 and infer_in_dim(SHIPPED, 512) == 13
    # (80-d input: F' = 6 bins cover 78 of the 80 dims -- the last 2 never reach the conv, seq2seq.py:52 / SURVEY 0)


def test_optional_model_features_build_their_parameter_sets():
    """rnn_config.ln / linear_proj / n_attn / feed_attn and cnn_config.bn = false (seq2seq.py:43-57, 81-121) are accepted; the parameter
    set follows the reference's links (names as chainer.serializers writes them) and matches the oracle's, value for value."""
    from ast_amd.params import init_values, param_shapes
    from ast_amd.seq2seq import SpeechEncoderDecoder
    from oracle import ast_ref as R
    cfg = json.loads(json.dumps(SHIPPED))
    cfg["rnn_config"].update(ln=True, n_attn=3, feed_attn=False, dec_vocab_size=40)
    cfg["cnn_config"]["bn"] = False
    m = SpeechEncoderDecoder(-1, cfg)
    assert m.rnn_ln and m.n_attn == 3 and not m.feed_attn and not m.cnn_bn and m.paths()["options"]
    train, persist = param_shapes(cfg, 13, 40)
    H, E = cfg["rnn_config"]["hidden_units"], cfg["rnn_config"]["embedding_units"]
    assert train["L2_rev_enc_ln/gamma"] == (H // 2,) and train["L1_dec_ln/beta"] == (H,) and train["attn_Wa2/W"] == (H, H)
    assert train["context/W"] == (cfg["rnn_config"]["attn_units"], 4 * H) and train["L0_dec/upward/W"] == (4 * H, E)
    assert train["CNN_1/b"] == (cfg["cnn_config"]["cnn_layers"][1]["out_channels"],) and not persist
    proj = json.loads(json.dumps(SHIPPED))
    proj["rnn_config"].update(linear_proj=True, dec_vocab_size=40)
    tp, pp = param_shapes(proj, 13, 40)
    assert tp["enc_proj1/W"] == (H, H) and "enc_proj2/W" not in tp and tp["L1_enc/upward/W"] == (4 * (H // 2), H) and "enc_proj0_bn/avg_var" in pp
    for c in (cfg, proj):
        a, b = init_values(c, 13, 40, seed=3), R.init_params(c, 13, 40, seed=3)
        assert set(a) == set(b) and all(np.array_equal(a[k], b[k]) for k in a)
    bad = json.loads(json.dumps(SHIPPED))
    bad["rnn_config"]["n_attn"] = 9
    with pytest.raises(ValueError):
        SpeechEncoderDecoder(-1, bad)


def test_compute_path_fails_loudly_without_gpu():
    import torch
    from ast_amd.seq2seq import SpeechEncoderDecoder
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = json.loads(json.dumps(SHIPPED))
    cfg["rnn_config"]["dec_vocab_size"] = 20
    m = SpeechEncoderDecoder(-1, cfg)
    with pytest.raises(RuntimeError):
        m.forward_loss(np.zeros((1, 20, 13), np.float32), np.array([[1, 2]], np.int32), 1.0)


def test_grad_buckets_tile_the_arena_of_every_config():
    """The overlapped data-parallel exchange cuts the flat gradient arena into CNN / encoder / decoder ranges: they must be
    contiguous and cover it for the benchmark model and for the shipped 3-layer-decoder variants."""
    import copy
    import types
    import torch
    import bench
    from ast_amd import dist as adist
    from ast_amd.params import ParamArena, param_shapes
    for enc_layers, dec_layers, feat in [(3, 1, 80), (3, 3, 13), (1, 2, 40)]:
        cfg = copy.deepcopy(bench.MODEL_CFG)
        cfg["rnn_config"]["enc_layers"] = enc_layers
        cfg["rnn_config"]["dec_layers"] = dec_layers
        train, _ = param_shapes(cfg, feat)
        arena = ParamArena(train, torch.device("cpu"))
        gb = adist.make_grad_buckets(types.SimpleNamespace(arena=arena))
        spans = sorted(gb.ranges.values())
        assert spans[0][0] == 0 and spans[-1][1] == arena.size
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert set(gb.ranges) == {"cnn", "enc", "dec"}


def test_bleu_matches_hand_computation(tmp_path):
    """ast_amd.eval restates nltk's corpus_bleu + SmoothingFunction.method2 (eval.py:30-38 of the reference)."""
    import math
    from ast_amd.eval import Eval, corpus_bleu
    ref = "the cat is on the mat".split()
    assert abs(corpus_bleu([[ref]], [ref]) - 1.0) < 1e-12                      # identical: every p_n = 1 (smoothed (k+1)/(k+1)), BP = 1
    assert corpus_bleu([[ref]], ["dogs bark loudly".split()]) == 0.0            # no unigram match
    # hypothesis "the the the cat" against two references: clipped unigrams: the<=2, cat<=1 -> 3/4; bigrams: "the cat" 1/3;
    # trigrams 0/2, 4-grams 0/1; method2 -> 3/4, 2/4, 1/3, 1/2; closest reference length to 4 is 5 (|6-4| > |5-4|) -> BP = exp(1 - 5/4)
    hyp = "the the the cat".split()
    refs = [ref, "there is the cat here".split()]
    want = math.exp(1 - 5 / 4) * math.exp(0.25 * (math.log(3 / 4) + math.log(2 / 4) + math.log(1 / 3) + math.log(1 / 2)))
    assert abs(corpus_bleu([refs], [hyp]) - want) < 1e-12
    # corpus level = micro-average over segments, not the mean of sentence scores
    two = corpus_bleu([[ref], refs], [ref, hyp])
    assert corpus_bleu([refs], [hyp]) < two < 1.0
    # the Eval class reads the reference's file layout
    (tmp_path / "eval.ids").write_text("u1\nu2\n")
    (tmp_path / "ref.en0").write_text("the cat is on the mat\nthere is the cat here\n")
    (tmp_path / "ref.en1").write_text("a cat sits on the mat\nthe cat is here\n")
    ev = Eval(str(tmp_path), 2)
    assert len(ev.refs) == 2 and ev.refs[0][1] == "a cat sits on the mat".split()
    assert abs(ev.calc_bleu({"u1": ref, "u2": "the cat is here".split()}) - 1.0) < 1e-12
    ev.write_to_file({"u1": ref, "u2": hyp}, str(tmp_path / "out.txt"))
    assert (tmp_path / "out.txt").read_text() == "the cat is on the mat\nthe the the cat\n"


def _write_corpus(tmp_path, nested):
    """A miniature corpus in the reference's on-disk schemas: info {set: {utt: {'sp': frames}}}, map {set: {utt: {dec_key: [bytes]}}},
    vocab {dec_key: {'w2i', 'i2w', 'freq'}}, per-utterance .npy files (optionally under <speaker>/ sub-directories)."""
    import numpy as np
    rng = np.random.default_rng(7)
    words = [b"_PAD", b"_GO", b"_EOS", b"_UNK", b"hel@@", b"lo", b"wor@@", b"ld", b"a"]
    vocab = {"bpe_w": {"w2i": {w: i for i, w in enumerate(words)}, "i2w": {i: w for i, w in enumerate(words)}, "freq": {}}}
    info, mp, speech = {}, {}, {}
    for set_key, n in (("fisher_train", 11), ("fisher_dev", 3)):
        info[set_key], mp[set_key], speech[set_key] = {}, {}, {}
        for i in range(n):
            utt = "spk{0:d}_{1:03d}".format(i % 2, i)
            t = int(rng.integers(30, 330))
            x = rng.standard_normal((t, 5)).astype(np.float32)
            info[set_key][utt] = {"sp": t}
            mp[set_key][utt] = {"bpe_w": [words[int(j)] for j in rng.integers(4, 9, size=int(rng.integers(1, 9)))] + ([b"zzz"] if i == 0 else [])}
            speech[set_key][utt] = x
            d = tmp_path / "speech" / set_key / ("spk{0:d}".format(i % 2) if nested else "")
            os.makedirs(d, exist_ok=True)
            np.save(d / (utt + ".npy"), x)
    for name, obj in (("info.dict", info), ("map.dict", mp), ("vocab.dict", vocab), ("speech.blob", speech)):
        pickle.dump(obj, open(tmp_path / name, "wb"))
    return info, mp, speech


@pytest.mark.parametrize("kind", ["fisher-flat", "fisher-nested", "globalphone"])
def test_file_loaders_follow_the_reference_schemas(tmp_path, kind):
    """dataloader.py:95-164 / :185-297: per-utterance .npy files (flat or <prefix>/<utt>.npy) and the pickled blob; targets through
    w2i with UNK, GO ... EOS, truncation to max_pred; zero padding; get_hyps joins BPE pieces."""
    import numpy as np
    from ast_amd.dataloader import SYMBOLS, FisherDataLoader, GlobalPhoneDataLoader
    info, mp, speech = _write_corpus(tmp_path, nested=(kind == "fisher-nested"))
    data = {"map_path": str(tmp_path / "map.dict"), "vocab_path": str(tmp_path / "vocab.dict"), "info_path": str(tmp_path / "info.dict"),
            "speech_path": str(tmp_path / ("speech.blob" if kind == "globalphone" else "speech")), "buckets_num": 4, "buckets_width": 80,
            "train_scale": 1, "dec_key": "bpe_w", "max_pred": 6, "zero_input": 0.0}
    dl = (GlobalPhoneDataLoader if kind == "globalphone" else FisherDataLoader)(data, str(tmp_path), -1)
    assert os.path.exists(tmp_path / "buckets_sp.dict") and dl.n_utts == {"fisher_train": 11, "fisher_dev": 3}
    random.seed(3)
    seen = []
    for batch in dl.get_batch(4, "fisher_train", train=True, labels=True):
        X, y = batch["X"].numpy(), batch["y"].numpy()
        for i, u in enumerate(batch["utts"]):
            x = speech["fisher_train"][u]
            assert np.array_equal(X[i, :len(x)], x) and not X[i, len(x):].any()
            ids = [dl.vocab["bpe_w"]["w2i"].get(w, SYMBOLS.UNK_ID) for w in mp["fisher_train"][u]["bpe_w"]]
            want = [SYMBOLS.GO_ID] + ids[:4] + [SYMBOLS.EOS_ID]
            assert list(y[i, :len(want)]) == want and not y[i, len(want):].any()
        assert len({min(info["fisher_train"][u]["sp"] // 80, 3) for u in batch["utts"]}) == 1
        seen += batch["utts"]
    assert sorted(seen) == sorted(info["fisher_train"])
    dl.data_cfg["max_pred"] = 64                                        # untruncated: the out-of-vocabulary word of utterance 0 maps to UNK
    assert int(dl._targets("spk0_000", "fisher_train")[-2]) == SYMBOLS.UNK_ID
    hyps = dl.get_hyps([("u", [4, 5, 6, 7, 8, 2]), ("v", [1, 8, 0])])
    assert hyps == {"u": ["hello", "world", "a"], "v": ["a"]}


def test_bench_starts_its_own_ranks_when_no_launcher_did(monkeypatch, capsys):
    """`python bench.py --gpus N` (N > 1) without WORLD_SIZE in the environment: bench.py starts torch.distributed.run as a CHILD process
    (one rank per GPU, rendezvous on 127.0.0.1, same arguments), relays rank 0's JSON line and exits with the child's code -- before
    anything touches HIP.  With WORLD_SIZE set (a launcher started us) or N = 1 nothing is spawned."""
    import subprocess
    import sys
    import types
    import bench
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.self_launch_command(8, argv, port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    k = cmd.index(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    assert cmd[k + 1:] == argv
    auto = bench.self_launch_command(2, argv)          # a free port is picked when none is given
    assert 1024 < int(auto[auto.index("--master-port") + 1]) < 65536

    calls = []

    class FakeProc:
        def __init__(self, cmd, **kw):
            calls.append((cmd, kw))
            self.stdout = iter(["rank 1 chatter\n", '{"metric": "speech frames/s (train step)", "value": 1.0, "n_gpus": 8}\n'])

        def wait(self):
            return 0
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = types.SimpleNamespace(gpus=8)
    with pytest.raises(SystemExit) as e:
        bench.maybe_self_launch(args, argv)
    assert e.value.code == 0 and len(calls) == 1
    assert calls[0][0][-len(argv):] == argv and "env" in calls[0][1] and calls[0][1]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert json.loads(out.out.strip())["n_gpus"] == 8 and "chatter" in out.err
    # a launcher already started us, or one GPU: no child
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.maybe_self_launch(args, argv)
    monkeypatch.delenv("WORLD_SIZE")
    bench.maybe_self_launch(types.SimpleNamespace(gpus=1), ["--gpus", "1"])
    assert len(calls) == 1


# ---------------------------------------------------------------------------------------------------------------------------------
# oracle/loader_ref.py: the reference's batch construction restated with injectable draws
def test_loader_oracle_known_answers():
    """Hand-derived: the bucket rule, frame zeroing with replacement, targets [GO] + ids[:max_pred-2] + [EOS] with UNK, truncation to
    (num_b + 1) * width_b frames, zero padding, and the order of the `random` calls (one shuffle per bucket, then the batch list)."""
    from oracle import loader_ref as LR
    info = {"a": {"sp": 5}, "b": {"sp": 79}, "c": {"sp": 80}, "d": {"sp": 400}, "e": {"sp": 239}}
    bk = LR.create_buckets(info, 3, 80, "sp", 1, "haha")
    assert bk["buckets"] == [["a", "b"], ["c"], ["d", "e"]]                       # min(frames // 80, 2)
    x = np.arange(20, dtype=np.float32).reshape(10, 2) + 1
    out = LR.drop_frames(x, 0.35, choice=lambda n, k: [2, 2, 9][:k])               # int(.35 * 10) = 3 draws, one repeated
    assert np.array_equal(np.where((out == 0).all(1))[0], [2, 9]) and np.array_equal(out[[0, 1, 3]], x[[0, 1, 3]])
    assert LR.drop_frames(x, 0.05) is x                                           # int(.5) = 0: untouched

    calls = []

    class Scripted:                       # reverses every list it is asked to shuffle, and records what it saw
        def shuffle(self, lst):
            calls.append(list(lst) if isinstance(lst[0], str) else "batches")
            lst.reverse()
    speech = {u: np.full((info[u]["sp"], 2), i + 1, dtype=np.float32) for i, u in enumerate(info)}
    vocab = {"w": {"w2i": {b"x": 4, b"y": 5}}}
    mp = {"train": {u: {"w": [b"x", b"y", b"q", b"x", b"y", b"x"][:n]} for u, n in zip(info, (1, 6, 3, 2, 4))}}
    ld = LR.RefLoader({"zero_input": 0.0, "dec_key": "w", "max_pred": 5}, {"train": bk}, vocab, mp, lambda u, k: speech[u], pyrandom=Scripted())
    got = list(ld.get_batch(2, "train", True, labels=True))
    assert calls == [["a", "b"], ["c"], ["d", "e"], "batches"]
    assert [b["utts"] for b in got] == [["e", "d"], ["c"], ["b", "a"]]           # buckets reversed inside, batch list reversed
    assert got[0]["X"].shape == (2, 320, 2)                                      # d: 400 frames truncated to (3 + 1) * 80
    assert (got[0]["X"][0, :239] == 5).all() and not got[0]["X"][0, 239:].any() and (got[0]["X"][1] == 4).all()
    assert got[2]["y"].tolist() == [[1, 4, 5, 3, 2], [1, 4, 2, 0, 0]]            # b: 6 words -> 3 kept (max_pred - 2), q -> UNK; a padded
    assert got[2]["y"].dtype == np.int32 and got[0]["X"].dtype == np.float32


@pytest.mark.parametrize("kind", ["fisher-flat", "globalphone"])
def test_host_loaders_reproduce_the_loader_oracle(tmp_path, kind):
    """The product's file loaders (host path) against the restated reference over two epochs, same seeds for Python's `random` and for
    NumPy's global RNG (frame zeroing, zero_input = 0.1): identical batch order, utterances, zeroed frames, padding and targets."""
    from ast_amd.dataloader import FisherDataLoader, GlobalPhoneDataLoader
    from oracle import loader_ref as LR
    info, mp, speech = _write_corpus(tmp_path, nested=False)
    data = {"map_path": str(tmp_path / "map.dict"), "vocab_path": str(tmp_path / "vocab.dict"), "info_path": str(tmp_path / "info.dict"),
            "speech_path": str(tmp_path / ("speech.blob" if kind == "globalphone" else "speech")), "buckets_num": 3, "buckets_width": 80,
            "train_scale": 1, "dec_key": "bpe_w", "max_pred": 6, "zero_input": 0.1}
    dl = (GlobalPhoneDataLoader if kind == "globalphone" else FisherDataLoader)(data, str(tmp_path), -1)
    vocab = pickle.load(open(tmp_path / "vocab.dict", "rb"))
    buckets = {k: LR.create_buckets(info[k], 3, 80, "sp", 1, "haha") for k in info}
    assert buckets["fisher_train"]["buckets"] == dl.buckets["fisher_train"]["buckets"]
    ref = LR.RefLoader(data, buckets, vocab, mp, lambda u, k: speech[k][u])
    for set_key, train in (("fisher_train", True), ("fisher_dev", False)):
        random.seed("seed-ast-20h")
        np.random.seed(5)
        want = [b for _ in range(2) for b in ref.get_batch(4, set_key, train, labels=True)]
        random.seed("seed-ast-20h")
        np.random.seed(5)
        got = [b for _ in range(2) for b in dl.get_batch(4, set_key, train=train, labels=True)]
        assert len(got) == len(want) > 1
        zeroed = 0
        for w, g in zip(want, got):
            assert w["utts"] == g["utts"]
            assert np.array_equal(w["X"], g["X"].numpy()) and np.array_equal(w["y"], g["y"].numpy())
            zeroed += sum(int((w["X"][i, :info[set_key][u]["sp"]] == 0).all(1).sum()) for i, u in enumerate(w["utts"]))
        assert (zeroed > 0) == train                                               # only "train" sets lose frames (dataloader.py:105)


def test_per_bucket_batch_sizes_follow_the_old_paths_create_batches(tmp_path):
    """nmt_run.py:406-447: batch_size = {'max','med','min'} by the bucket's third, bucket order shuffled first (or ascending for a
    curriculum, then nothing else is shuffled between buckets) -- the product's batch_plan against the restated create_batches on the
    same `random` stream."""
    from ast_amd.dataloader import SyntheticDataLoader
    from oracle import loader_ref as LR
    data = {"dataloader": "synthetic", "vocab_size": 31, "feat_dim": 5, "n_utts": {"syn_train": 90}, "frames": [5, 470], "targets": [1, 9],
            "buckets_num": 6, "buckets_width": 80, "max_pred": 12, "zero_input": 0.0, "train_scale": 1, "dec_key": "bpe_w"}
    sizes = {"max": 7, "med": 4, "min": 2}
    for curriculum in (False, True):
        dl = SyntheticDataLoader(data, str(tmp_path), -1)
        ref = {"buckets": [list(b) for b in dl.buckets["syn_train"]["buckets"]], "num_b": 6, "width_b": 80}
        random.seed(12)
        plan = dl.batch_plan(dict(sizes, curriculum=curriculum), "syn_train")
        random.seed(12)
        want, total = LR.create_batches(ref, sizes, curriculum=curriculum)
        assert total == 90 and [u for u, _ in plan] == [u for u, _ in want]
        assert [w for _, w in plan] == [(b + 1) * 80 for _, b in want]
        by_third = {0: 7, 1: 7, 2: 4, 3: 4, 4: 2, 5: 2}
        assert all(len(u) <= by_third[b] for u, b in want) and any(len(u) == 7 for u, _ in want)
        if curriculum:
            assert [b for _, b in want] == sorted(b for _, b in want)


@pytest.mark.parametrize("T,D,kt,kf,st,pt", [(800, 80, 9, 13, 2, 4), (37, 26, 9, 13, 2, 4), (50, 80, 5, 7, 2, 1)])
def test_layer0_patch_is_a_contiguous_window_of_the_frequency_blocked_input(T, D, kt, kf, st, pt):
    """The identity the direct layer-0 convolution (conv.hip: k_conv0_fwd_x3) and its weight gradient rest on, restated in NumPy: when the
    frequency stride equals the frequency kernel (seq2seq.py:43-57: ksize (9, 13), stride (2, 13)), the im2col patch of output step t1 of
    frequency block f is the kt * J CONTIGUOUS elements from flat offset st * J * t1 of that block's [pt + T + pt][J] image (columns >= kf
    zero), so W[c][i][j] placed at i * J + j reproduces the convolution -- for the kernel's LDS image (J = 20) and for XF (J = 14) alike."""
    rng = np.random.default_rng(T + kt)
    F, T1, C = (D - kf) // kf + 1, (T + 2 * pt - kt) // st + 1, 5
    X, W = rng.standard_normal((T, D)), rng.standard_normal((C, kt, kf))
    # reference: the convolution as the reference writes it (time padding pt, no frequency padding)
    Xp = np.zeros((T + 2 * pt, D)); Xp[pt:pt + T] = X
    ref = np.zeros((F, T1, C))
    for f in range(F):
        for t1 in range(T1):
            ref[f, t1] = np.einsum("cij,ij->c", W, Xp[st * t1:st * t1 + kt, kf * f:kf * f + kf])
    for J in (20, (kf + 1) & ~1):
        rows = (T + 2 * pt + 1) & ~1
        Wj = np.zeros((C, kt * J)); Wj.reshape(C, kt, J)[:, :, :kf] = W
        for f in range(F):
            img = np.zeros((rows + kt, J)); img[pt:pt + T, :kf] = X[:, kf * f:kf * f + kf]      # the frequency-blocked, time-padded image
            flat = img.ravel()
            got = np.stack([Wj @ flat[st * J * t1:st * J * t1 + kt * J] for t1 in range(T1)])
            np.testing.assert_allclose(got, ref[f], rtol=0, atol=1e-12)
