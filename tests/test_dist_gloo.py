"""world_size-2 gloo tests of the data-parallel plumbing (ast_amd.dist): the flat-gradient mean all-reduce, the
identical teacher-forcing stream on every rank, and that averaging per-replica gradients of the reference's loss
(mean over the LOCAL batch, quirk Q6) reproduces the single-process gradient of the global batch for everything
that does not go through BatchNorm statistics, and the exchange step of the global-batch BatchNorm option
(its numerics are checked on the GPU: tests/test_gpu_model.py::test_sync_batchnorm_...)."""
import os
import random
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ast_amd import dist as adist
    from ast_amd.params import ParamArena
    adist.init("gloo")
    assert adist.is_distributed() and adist.rank() == rank and adist.world_size() == world
    # 1) flat arena all-reduce = mean over ranks, in place
    arena = ParamArena({"a/W": (5, 3), "b/b": (6,)}, torch.device("cpu"))
    arena.grad.copy_(torch.arange(arena.size, dtype=torch.float32) * (rank + 1))
    adist.allreduce_grads(arena)
    expect = torch.arange(arena.size, dtype=torch.float32) * (1 + world) / 2
    ok1 = bool(torch.allclose(arena.grad, expect))
    # 1b) bucketed, overlapped exchange (GradBuckets) == the flat all-reduce, whatever the launch order
    from ast_amd.dist import GradBuckets
    shapes = {"CNN_0/W": (3, 1, 2, 2), "CNN_0_bn/gamma": (3,), "L0_enc/upward/W": (8, 5), "L0_rev_enc/lateral/W": (8, 2),
              "attn_Wa/W": (4, 4), "out/b": (7,)}
    ar2 = ParamArena(shapes, torch.device("cpu"))
    base = torch.arange(ar2.size, dtype=torch.float32) + 1
    ar2.grad.copy_(base * (rank + 2))
    gb = GradBuckets(ar2, {"cnn": [n for n in shapes if n.startswith("CNN_")], "enc": [n for n in shapes if "_enc" in n],
                           "dec": ["attn_Wa/W", "out/b"]})
    gb.launch("dec")
    gb.launch("enc")
    gb.finish()                      # launches "cnn" itself, waits, scales
    ok1 = ok1 and bool(torch.allclose(ar2.grad, base * sum(r + 2 for r in range(world)) / world))
    # 2) same seed string -> same teacher-forcing coins on every rank (quirk Q4 under DP)
    random.seed("seed-ast-20h")
    coins = torch.tensor([random.random() for _ in range(8)], dtype=torch.float64)
    gathered = [torch.zeros_like(coins) for _ in range(world)]
    td.all_gather(gathered, coins)
    ok2 = all(bool(torch.equal(g, coins)) for g in gathered)
    # 3) decoder-only DP equivalence with the oracle: mean of per-shard gradients == global-batch gradient
    from oracle.ast_ref_torch import decoder_torch
    rng = np.random.default_rng(0)
    B, L, T, H, E, A, V = 4, 5, 6, 8, 4, 8, 11
    cfg = {"rnn_config": {"dec_layers": 1, "attn_units": A}}
    P = {"embed_dec/W": rng.standard_normal((V, E)), "attn_Wa/W": rng.standard_normal((H, H)) * .3, "attn_Wa/b": np.zeros(H),
         "context/W": rng.standard_normal((A, 2 * H)) * .3, "context/b": np.zeros(A), "out/W": rng.standard_normal((V, A)) * .3,
         "out/b": np.zeros(V), "L0_dec/upward/W": rng.standard_normal((4 * H, E + A)) * .3, "L0_dec/upward/b": np.zeros(4 * H),
         "L0_dec/lateral/W": rng.standard_normal((4 * H, H)) * .3}
    enc, c0, h0 = rng.standard_normal((B, T, H)), rng.standard_normal((1, B, H)), rng.standard_normal((1, B, H)) * .5
    y = np.array([[1, 5, 6, 7, 2], [1, 4, 2, 0, 0], [1, 9, 8, 2, 0], [1, 3, 3, 3, 2]])
    flags = [1, 1, 1, 1]

    def grads(rows):
        Pt = {k: torch.tensor(v, requires_grad=True) for k, v in P.items()}
        loss, _ = decoder_torch(cfg, Pt, torch.tensor(enc[rows]), torch.tensor(c0[:, rows]), torch.tensor(h0[:, rows]), y[rows], flags, V)
        loss.backward()
        return torch.cat([Pt[k].grad.reshape(-1) for k in sorted(Pt)])
    rows = adist.shard_rows(B, rank, world)            # strided, like the loader's utts[rank::world]
    assert rows == list(range(B))[rank::world] and adist.shard_rows(B + 1, rank, world) == rows
    local = grads(rows).float()
    adist.allreduce_flat(local)
    ok3 = bool(torch.allclose(local.double(), grads(slice(0, B)), rtol=1e-5, atol=1e-7))
    # 4) BatchNorm statistics exchange (StatExchange): called the way the C library calls it -- through the ctypes callback, with
    #    the address of n float64 sums inside the workspace tensor -- it must leave the sum over ranks in place, and refuse
    #    addresses outside the bound workspace
    ws = torch.zeros(4096, dtype=torch.uint8)
    stats = ws[1024:1024 + 8 * 6].view(torch.float64)
    stats.copy_(torch.arange(6, dtype=torch.float64) + 10 * rank)
    sx = adist.StatExchange().bind(ws)
    rc = sx.callback(None, ws.data_ptr() + 1024, 6, None)
    ok4 = rc == 0 and sx.world == world and bool(torch.equal(stats, world * torch.arange(6, dtype=torch.float64) + 10 * sum(range(world))))
    ok4 = ok4 and bool((ws[:1024] == 0).all()) and bool((ws[1024 + 48:] == 0).all())
    ok4 = ok4 and sx.callback(None, ws.data_ptr() + 4090, 6, None) == -1 and isinstance(sx.error, RuntimeError)
    q.put((rank, ok1, ok2, ok3, ok4))
    td.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3, ok4 in res:
        assert ok1, f"rank {rank}: all-reduce mean wrong"
        assert ok2, f"rank {rank}: teacher-forcing streams differ"
        assert ok3, f"rank {rank}: averaged shard gradients != global-batch gradient"
        assert ok4, f"rank {rank}: BatchNorm statistics exchange wrong"


def _abort_worker(rank, world, port):
    """Rank 1 reports a persistent-kernel time-out in its status word; the word rides behind the last gradient range, so after the
    exchange BOTH ranks see it, skip the step and raise -- nobody is left waiting in the next collective."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ast_amd import dist as adist
    from ast_amd.dist import GradBuckets
    from ast_amd.params import ParamArena
    adist.init("gloo")
    shapes = {"CNN_0/W": (3, 1, 2, 2), "L0_enc/upward/W": (8, 5), "attn_Wa/W": (4, 4), "out/b": (7,)}
    arena = ParamArena(shapes, torch.device("cpu"))
    groups = {"cnn": ["CNN_0/W"], "enc": ["L0_enc/upward/W"], "dec": ["attn_Wa/W", "out/b"]}
    base = torch.arange(arena.size, dtype=torch.float32) + 1
    status = {"v": 0.0}
    gb = GradBuckets(arena, groups, defer_scale=True, status_fn=lambda tail: tail.fill_(status["v"]))
    # step 1: healthy on both ranks -> the tail sums to 0, gradients are the plain sums, nothing raises
    arena.grad.copy_(base * (rank + 1))
    gb.launch("dec"); gb.launch("enc")
    scale = gb.finish()
    assert scale == 1.0 / world and float(gb.status_sum[0]) == 0.0
    assert torch.allclose(arena.grad, base * sum(r + 1 for r in range(world)))
    adist.raise_if_any_rank_aborted(gb)
    # step 2: rank 1's decoder forward (bit 4) timed out
    status["v"] = 4.0 if rank == 1 else 0.0
    arena.grad.copy_(base)
    gb.launch("dec"); gb.launch("enc")
    gb.finish()
    assert float(gb.status_sum[0]) == 4.0, (rank, float(gb.status_sum[0]))       # every rank holds the SUM
    assert torch.allclose(arena.grad, base * world)                               # the tail did not disturb the gradients
    adist.raise_if_any_rank_aborted(gb, "gloo test step 2")                       # raises AstkError on BOTH ranks -> exit code 1
    td.destroy_process_group()                                                    # not reached


def test_abort_word_reaches_every_rank_through_the_gradient_all_reduce():
    """Round-3 review item 6: a persistent-kernel time-out on ONE rank must stop ALL ranks in the same step.  Two gloo ranks; rank 1
    reports a fake time-out; both must exit non-zero within seconds (before: rank 0 trained on and then waited in the next all-reduce
    until the 120-minute watchdog)."""
    import time
    ctx = mp.get_context("spawn")
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_abort_worker, args=(r, 2, port)) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert time.time() - t0 < 120
    assert [p.exitcode for p in procs] == [1, 1], [p.exitcode for p in procs]


def _pipelined_abort_worker(rank, world, port, q):
    """NN.train_epoch's loop shape: the [loss, status] pair of step N is read back AFTER step N+1 has been enqueued.  Rank 1 reports a
    time-out in step 2.  The merged word must land in the pair of step 2 on BOTH ranks (GradBuckets.status_dest), so that both raise at
    the read-back of step 2 with the same collectives enqueued (steps 0..3) -- nobody is left waiting in the all-reduce of step 4."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ast_amd import dist as adist
    from ast_amd.dist import GradBuckets
    from ast_amd.params import ParamArena
    from ast_amd.seq2seq import raise_if_aborted
    from ast_amd._lib import AstkError
    adist.init("gloo")
    shapes = {"CNN_0/W": (3, 1, 2, 2), "L0_enc/upward/W": (8, 5), "attn_Wa/W": (4, 4), "out/b": (7,)}
    arena = ParamArena(shapes, torch.device("cpu"))
    groups = {"cnn": ["CNN_0/W"], "enc": ["L0_enc/upward/W"], "dec": ["attn_Wa/W", "out/b"]}
    status = {"v": 0.0}
    gb = GradBuckets(arena, groups, defer_scale=True, status_fn=lambda tail: tail.fill_(status["v"]))
    pending, enqueued, raised_at = None, 0, None
    try:
        for step in range(7):
            status["v"] = 4.0 if (rank == 1 and step == 2) else 0.0          # rank 1's decoder forward times out in step 2
            pair = torch.tensor([float(step + 1), status["v"]])              # the pair's own snapshot, taken before the exchange
            gb.status_dest = pair[1:2]                                       # (what SpeechEncoderDecoder._backward sets)
            arena.grad.fill_(1.0)
            gb.launch("dec"); gb.launch("enc")
            gb.finish()                                                      # optimizer.update(): the step's last collective + the merge
            enqueued += 1
            cur = (pair.clone(), step)
            if pending is not None:
                raised_at = pending[1]
                raise_if_aborted(pending[0].tolist()[1], "gloo pipeline test")
                raised_at = None
            pending = cur
    except AstkError:
        q.put((rank, enqueued, raised_at))
        td.destroy_process_group()
        return
    q.put((rank, enqueued, None))
    td.destroy_process_group()


def test_pipelined_loss_readback_raises_on_every_rank_at_the_same_step():
    """with the loss of step N read back after step N+1 has been enqueued, the merged abort word has to be in
    the pair of step N on every rank; otherwise the failing rank stops one step before the healthy ones, which then wait in an
    all-reduce nobody joins."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipelined_abort_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, 4, 2), (1, 4, 2)], res          # both ranks: raised at the read-back of step 2, with steps 0..3 enqueued


def _random_out_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ast_amd import dist as adist
    from ast_amd.seq2seq import draw_flags_and_targets
    adist.init("gloo")
    V, B, L = 23, 3, 9
    rng = np.random.default_rng(100 + rank)                     # DIFFERENT shards: different numbers of targets >= 4 per rank
    y = np.zeros((B, L), dtype=np.int32)
    for b in range(B):
        n = int(rng.integers(4, L + 1))
        y[b, 0], y[b, 1:n - 1], y[b, n - 1] = 1, rng.integers(4, V, size=n - 2), 2
    random.seed("seed-ast-20h")
    flags, tg = draw_flags_and_targets(y, 0.8, 0.3, V, lambda lo, hi: hi - 1)          # replaced entries become V - 1 (clamped from V)
    state = random.getstate()
    after = random.random()
    q.put((rank, flags, y.tolist(), tg.tolist(), list(state[1][-8:]), after))      # (tail of the Mersenne state + its index)
    td.destroy_process_group()


def test_random_out_draws_keep_the_shared_random_stream_identical_on_every_rank():
    """with random_out > 0 the number of draws from the seeded Python `random` stream depended on each rank's own
    rows, so the ranks' streams -- which also shuffle next epoch's batches -- drifted apart.  Now every rank makes the draws of the whole
    global batch in the unsharded row order and keeps its own rows' replacements: same flags, same stream state, and together the
    ranks' scored targets are exactly what ONE process computes on the concatenated batch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    world = 2
    procs = [ctx.Process(target=_random_out_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, f0, y0, t0, s0, a0), (_, f1, y1, t1, s1, a1) = res
    assert f0 == f1 and s0 == s1 and a0 == a1, "the ranks' random streams drifted apart"
    assert t0 != y0 or t1 != y1, "nothing was replaced: the test draws nothing"
    # one process on the concatenated batch (global row b * world + r = row b of rank r)
    sys.path.insert(0, ROOT)
    from ast_amd.seq2seq import draw_flags_and_targets
    yg = np.empty((2 * len(y0), len(y0[0])), dtype=np.int32)
    yg[0::2], yg[1::2] = np.asarray(y0), np.asarray(y1)
    random.seed("seed-ast-20h")
    fg, tg = draw_flags_and_targets(yg, 0.8, 0.3, 23, lambda lo, hi: hi - 1, rank=0, world=1)
    assert fg == f0 and random.random() == a0
    assert tg[0::2].tolist() == t0 and tg[1::2].tolist() == t1


def test_init_caps_the_rccl_channels_like_bench_does(monkeypatch):
    """the channel cap used to be applied by bench.py only; train.py -> dist.init() ran uncapped.  init("nccl")
    now applies it itself (checked here without a process group: WORLD_SIZE 2, torch.distributed stubbed)."""
    from ast_amd import dist as adist
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("LOCAL_RANK", "0")
    called = {}
    monkeypatch.setattr(adist.td, "is_initialized", lambda: False)
    monkeypatch.setattr(adist.td, "init_process_group", lambda **kw: called.update(kw))
    monkeypatch.setattr(adist.torch.cuda, "set_device", lambda d: called.update(device=d))
    adist.init("nccl")
    assert called["backend"] == "nccl" and os.environ["NCCL_MAX_NCHANNELS"] == "32" and adist.channel_cap == 32
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    adist.init("nccl", recurrence_cus=240)
    assert os.environ["NCCL_MAX_NCHANNELS"] == "16"
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    adist.init("gloo")
    assert "NCCL_MAX_NCHANNELS" not in os.environ                 # only the RCCL backend is capped


def test_rccl_channel_cap_leaves_the_recurrence_grid_its_cus(monkeypatch):
    """ast_amd.dist.reserve_cus_for_recurrence: RCCL kernels hold one CU per channel until every peer has arrived; the persistent
    recurrence grids need their workgroups resident at once.  The cap keeps channels + grid within the device; an explicit, too large
    NCCL_MAX_NCHANNELS is refused."""
    from ast_amd import dist as adist
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    assert adist.reserve_cus_for_recurrence(192) == 32 and os.environ["NCCL_MAX_NCHANNELS"] == "32"
    monkeypatch.delenv("NCCL_MAX_NCHANNELS")
    assert adist.reserve_cus_for_recurrence(240) == 16
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "8")
    assert adist.reserve_cus_for_recurrence(192) == 8                 # an explicit smaller value stays
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "128")
    with pytest.raises(RuntimeError):
        adist.reserve_cus_for_recurrence(192)
