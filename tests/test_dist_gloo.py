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
