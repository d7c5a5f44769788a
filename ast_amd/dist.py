"""Data parallelism: one process per GPU (torchrun), replicas of the bucketed minibatch, ONE all-reduce of the flat
float32 gradient arena per step over RCCL/xGMI (backend "nccl" on ROCm), scaled by 1/world so that the per-replica
1/B_local cross-entropy means compose to the global 1/B mean.  All ranks seed Python's `random`
identically (nn.py:54), so the teacher-forcing flags agree.  BatchNorm uses per-replica statistics by default (what N independent
Chainer processes would do); `StatExchange` gives the global-batch statistics of one process on the concatenated batch.
CPU tests run the same code over gloo."""
import os

import torch
import torch.distributed as td


def is_distributed():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


# persistent grids of the shipped models: encoder recurrences (h/16) x ceil(B/16) x cells = 192 workgroups at batch 32, up to 224 for the
# widths the launchers accept beside a collective; the decoder loop takes every CU but never overlaps a collective (seq2seq.py)
DEFAULT_RECURRENCE_CUS = 224
channel_cap = None          # NCCL_MAX_NCHANNELS in force after init("nccl")


def init(backend=None, recurrence_cus=None):
    """Initialises torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  For the RCCL backend the channel
    count is capped FIRST (reserve_cus_for_recurrence) so that a collective waiting for a late peer cannot keep the next step's
    persistent recurrence grid from becoming resident: training (train.py -> NN) and bench.py run the same configuration.
    `recurrence_cus`: the largest persistent grid that can run beside a collective (default 224)."""
    global channel_cap
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or td.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        channel_cap = reserve_cus_for_recurrence(DEFAULT_RECURRENCE_CUS if recurrence_cus is None else recurrence_cus)
        torch.cuda.set_device(local_rank())
    # rank 0 alone runs the dev pass and writes checkpoints between epochs while the others wait at a barrier (train.py): the
    # collective watchdog must outlast a dev decode
    import datetime
    minutes = float(os.environ.get("ASTK_DIST_TIMEOUT_MIN", "120"))
    td.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=minutes))


def barrier():
    if is_distributed():
        td.barrier()


def reserve_cus_for_recurrence(need_cus, n_cu=256):
    """Call BEFORE init(): caps RCCL's channel count so that its kernels and the persistent recurrence grids fit the device together.
    An RCCL kernel occupies one CU per channel until every peer has arrived; the persistent encoder / decoder kernels need
    `need_cus` workgroups resident at once (one per CU; 192 for the shipped encoder at batch 32, 256 for the decoder loop, which
    therefore never overlaps a collective: ast_amd/seq2seq.py launches the all-reduces behind the recurrences).  With
    NCCL_MAX_NCHANNELS <= n_cu - need_cus a collective that is still waiting for a late peer when the NEXT step's encoder recurrence
    starts cannot keep it from becoming resident.  An explicit NCCL_MAX_NCHANNELS in the environment is respected (and checked).
    Returns the cap in force."""
    free = max(int(n_cu) - int(need_cus), 4)
    cur = os.environ.get("NCCL_MAX_NCHANNELS")
    if cur is None:
        os.environ["NCCL_MAX_NCHANNELS"] = str(min(free, 32))
    elif int(cur) > free:
        raise RuntimeError(f"NCCL_MAX_NCHANNELS={cur} leaves fewer than the {need_cus} CUs the persistent recurrence grid needs "
                           f"({n_cu} CUs): set it to {free} or less")
    return int(os.environ["NCCL_MAX_NCHANNELS"])


def allreduce_flat(buf):
    """In-place mean of one flat buffer over all ranks."""
    if not is_distributed():
        return buf
    td.all_reduce(buf, op=td.ReduceOp.SUM)
    buf.mul_(1.0 / world_size())
    return buf


def allreduce_grads(arena):
    return allreduce_flat(arena.grad)


class GradBuckets:
    """Overlapped gradient exchange: the flat gradient arena is cut into contiguous ranges that become final at different
    points of the backward pass (arena order = [CNN | encoder LSTMs | attention + decoder], backward order = decoder ->
    encoder -> CNN).  `launch(name)` starts an asynchronous all-reduce of one range on the communication stream as soon as
    its producer has been enqueued; `finish(arena)` (installed as the optimizer's `grad_sync`) launches whatever is left,
    waits for everything and applies the 1/world mean.  xGMI is point-to-point and a 53 MB all-reduce is ~1 ms of an ~9.5 ms
    step: behind the encoder / CNN backward it costs nothing.  Numerically identical to `allreduce_grads` (same sums)."""

    def __init__(self, arena, param_groups, defer_scale=False, status_fn=None):
        # param_groups: ordered {bucket name: [param names]}; every group must be one contiguous arena range.
        # defer_scale: finish() leaves the SUM in the arena and returns 1/world for the optimizer kernels to apply on the fly.
        # status_fn(tail): writes this rank's abort status (0 = healthy) into the one-float tensor `tail` on the current stream.  The
        #   tail sits directly behind the LAST arena range and rides in that range's all-reduce, so after the exchange every rank holds
        #   the SUM of all ranks' status words: a persistent kernel that timed out on one rank (its results are garbage) makes EVERY
        #   rank skip the update of that step and raise at the same loss read-back, instead of the healthy ranks training on and then
        #   sitting in the next all-reduce until the watchdog (round-3 review, item 6).  Default: the library's sticky status word
        #   (astk_persist_status_snapshot) on a GPU, nothing on the CPU.
        self.arena = arena
        self.defer_scale = defer_scale
        self.status_fn = status_fn if status_fn is not None else _library_status
        self.status_sum = arena.status_tail if getattr(arena, "status_tail", None) is not None else None
        # status_dest: the status word of the CURRENT step's [loss, status] pair (set by the model's backward).  The pair's own snapshot
        # is taken at the end of the backward, BEFORE finish() merges the peers' words: in the pipelined train loop (the loss of step N is
        # read back after step N+1 has been enqueued) the failing rank would then raise at the read-back of step N while the healthy ranks
        # first see bit 16 in the pair of step N+1, i.e. only after they have enqueued the all-reduce of step N+2 -- which the failing rank
        # never joins.  finish() therefore writes the MERGED word into the pair of the step it belongs to: every
        # rank raises at the read-back of step N with the same collectives enqueued.
        self.status_dest = None
        self.ranges = {}
        covered = []
        for gname, names in param_groups.items():
            if not names:
                continue
            spans = sorted(arena.range_of(n) for n in names)
            lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
            assert sum(n for _, n in spans) == hi - lo, f"bucket {gname} is not contiguous in the arena"
            self.ranges[gname] = (lo, hi)
            covered.append((lo, hi))
        covered.sort()
        assert covered and covered[0][0] == 0 and covered[-1][1] == arena.size and all(
            a[1] == b[0] for a, b in zip(covered, covered[1:])), "buckets must tile the arena"
        self.pending = {}

    def launch(self, name):
        if not is_distributed() or name in self.pending or name not in self.ranges:
            return
        lo, hi = self.ranges[name]
        buf = self.arena.grad[lo:hi]
        if hi == self.arena.size and self.status_sum is not None:
            # the range that ends the arena carries the status tail: [gradients | status | 3 pad floats] in one all-reduce
            self.status_fn(self.status_sum)
            buf = self.arena.grad_full[lo:hi + self.arena.TAIL]
        self.pending[name] = td.all_reduce(buf, op=td.ReduceOp.SUM, async_op=True)

    def finish(self, arena=None):
        if not is_distributed():
            self.pending = {}
            return
        for name in self.ranges:
            self.launch(name)
        for work in self.pending.values():
            work.wait()
        self.pending = {}
        if self.status_sum is not None and self.status_sum.is_cuda:
            # a non-zero SUM marks the step as aborted in this rank's sticky status word too (bit 16): the update kernels that follow
            # skip on every rank alike, and every rank's next loss read-back raises (seq2seq.raise_if_aborted)
            import ctypes as C
            from . import _lib
            stream = C.c_void_p(torch.cuda.current_stream(self.status_sum.device).cuda_stream)
            _lib.check(_lib.load().astk_persist_status_merge(C.c_void_p(self.status_sum.data_ptr()), stream))
            if self.status_dest is not None:        # the merged word into this step's [loss, status] pair (see __init__)
                _lib.check(_lib.load().astk_persist_status_snapshot(C.c_void_p(self.status_dest.data_ptr()), stream))
        elif self.status_sum is not None and self.status_dest is not None:
            # CPU (gloo tests): the library's sticky word does not exist here; the same rule on the tensors themselves -- a non-zero SUM sets
            # mask 16 ("reported by a peer rank") in the pair's status word, next to whatever this rank reported itself
            if float(self.status_sum[0]) != 0.0:
                self.status_dest[0] = float(int(self.status_dest[0]) | 16)
        if self.defer_scale:
            return 1.0 / world_size()
        self.arena.grad.mul_(1.0 / world_size())


def _library_status(tail):
    """This rank's sticky status word of the persistent kernels, as a float, into `tail` on the current stream (GPU only)."""
    if tail.is_cuda:
        import ctypes as C
        from . import _lib
        _lib.check(_lib.load().astk_persist_status_snapshot(C.c_void_p(tail.data_ptr()), C.c_void_p(torch.cuda.current_stream(tail.device).cuda_stream)))
    else:
        tail.zero_()


def raise_if_any_rank_aborted(buckets, where="train step"):
    """Host-side check of the all-reduced status word (synchronises; the GPU train loop does not need it -- the merged word travels
    through astk_persist_status_merge into the sticky word every rank reads next to its loss -- but CPU tests and callers without the
    loss read-back do): raises on EVERY rank when any rank reported an abort in the step just exchanged."""
    if buckets is None or buckets.status_sum is None:
        return
    v = float(buckets.status_sum[0])
    if v != 0.0:
        from . import _lib
        raise _lib.AstkError(f"{where}: a persistent kernel timed out on at least one rank (summed status word {v:g}); the step's update "
                             "was skipped on every rank")


def make_grad_buckets(model):
    """Buckets of a SpeechEncoderDecoder: CNN, encoder, decoder (+attention, embedding, output)."""
    groups = {"cnn": [], "enc": [], "dec": []}
    for name in model.arena.shapes:
        link = name.split("/")[0]
        # (L{i}_enc / L{i}_rev_enc, their _ln links and the enc_proj{i} links of the optional encoder variants are encoder parameters)
        is_enc = link.endswith(("_enc", "_enc_ln")) or link.startswith("enc_proj")
        groups["cnn" if link.startswith("CNN_") else "enc" if is_enc else "dec"].append(name)
    return GradBuckets(model.arena, groups, defer_scale=True)


def broadcast_params(arena, src=0):
    if is_distributed():
        td.broadcast(arena.data, src=src)


def shard_rows(n_rows, rank_, world):
    """Row indices of `rank_`'s shard of a bucketed batch -- the ONE sharding scheme of the data-parallel path, the same the loader
    applies to utterance lists (ast_amd/dataloader.py get_batch: `utts[rank::world]` after the batch has been cut to a multiple
    of the world size): strided rows rank, rank+world, ...; at most world-1 trailing rows sit the step out, so every rank
    gets the same number of rows (equal T => equal work, and the all-reduce never waits for an empty shard)."""
    usable = n_rows // world * world
    return list(range(rank_, usable, world))


class StatExchange:
    """The exchange step of data-parallel BatchNorm with global-batch statistics (include/astk.h, astk_conv_bn_relu_*_sync):
    the C library calls `self.callback(user, stat, n, stream)` between producing a layer's local per-channel sums (n float64
    values inside the CNN workspace tensor) and consuming them; the callback sums them over the ranks in place.  The
    all-reduce is enqueued behind the producing kernels (torch orders its communication stream after the current stream, which
    is the stream the library launches on) and the consumers are enqueued after it returns.  `reduce` can be replaced for tests."""

    def __init__(self, world=None, reduce=None):
        import ctypes as C
        self.world = world_size() if world is None else int(world)
        self.workspace = None     # the uint8 tensor the library was given as `ws` for the current call
        self.calls = 0
        self.reduce = reduce if reduce is not None else (lambda view: td.all_reduce(view, op=td.ReduceOp.SUM))
        self.error = None
        self._ctype = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
        self.callback = self._ctype(self._call)

    def view(self, ptr, n):
        """float64 view of n statistics at device address `ptr` inside the bound workspace tensor"""
        ws = self.workspace
        if ws is None:
            raise RuntimeError("StatExchange: no workspace bound")
        off = int(ptr) - ws.data_ptr()
        if off < 0 or off % 8 or off + 8 * n > ws.numel() * ws.element_size():
            raise RuntimeError(f"statistics buffer {ptr:#x}+{8 * n} is outside the bound workspace")
        return ws.view(torch.uint8)[off:off + 8 * n].view(torch.float64)

    def _call(self, user, stat, n, stream):
        try:                                   # exceptions cannot cross the C frame: report failure through the return code
            self.reduce(self.view(stat, n))
            self.calls += 1
            return 0
        except Exception as e:                 # noqa: BLE001
            self.error = e
            return -1

    def bind(self, workspace):
        self.workspace = workspace
        self.error = None
        return self
