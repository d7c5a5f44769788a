"""Exercises the data-parallel code path with the RCCL backend on ONE GPU (a process group of one rank): bucketed asynchronous
all-reduces launched from the backward pass, the side stream, the non-default compute stream and the optional BatchNorm
statistics exchange all run against the real backend; with one rank the result must equal the plain step."""
import copy, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # run as a script from tests/
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch, torch.distributed as td
import bench
from ast_amd import dist as adist, optimizers as O
from ast_amd.seq2seq import SpeechEncoderDecoder, using_config
from oracle.ast_ref import synth_batch
torch.cuda.set_device(0)
td.init_process_group("nccl", world_size=1, rank=0)
adist.is_distributed = lambda: True                      # world of one: the collectives are identities but go through RCCL
cfg = copy.deepcopy(bench.MODEL_CFG)
B, T, D, L, V = 32, 800, 80, 40, cfg["rnn_config"]["dec_vocab_size"]
X, y = synth_batch(B, T, D, L, V, 20)
X, y = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
def run(dp):
    m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
    m.rng_seed = 1234
    opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
    opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
    if dp:
        m.grad_buckets = adist.make_grad_buckets(m)
        opt.grad_sync = m.grad_buckets.finish
        m.stat_exchange = adist.StatExchange(world=1)
    random.seed("seed-ast-20h")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    losses = []
    with torch.cuda.stream(s):
        for _ in range(4):
            with using_config("train", True):
                l = m.forward_loss(X=X, y=y, teach_ratio=0.8, random_out=0, add_noise=0.25)
                m.cleargrads(); l.backward(); opt.update()
            losses.append(l.data.clone())
    torch.cuda.synchronize()
    return [float(v) for v in losses], m.arena.data.clone()
a, pa = run(False)
b, pb = run(True)
print("plain", a); print("dp   ", b)
assert all(abs(x - z) <= 1e-4 * abs(x) for x, z in zip(a, b)), "losses differ"
rel = float((pa - pb).abs().max() / pa.abs().max())
print("max parameter difference after 4 steps (relative):", rel)
assert rel < 1e-3
# ---- the abort word across ranks (round-3 review item 6), against the real backend: a status word that some rank reports as non-zero
# comes back from the all-reduce, astk_persist_status_merge marks THIS rank's sticky word (bit 16), the update kernels skip, and the next
# loss read-back raises.  (One rank: the "peer" is a status_fn that reports a decoder-forward time-out.)
from ast_amd import _lib
m = SpeechEncoderDecoder(0, cfg).materialize(D, seed=0)
opt = O.Adam(alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, amsgrad=True).setup(m)
opt.add_hook(O.WeightDecay(1e-4)); opt.add_hook(O.GradientClipping(2))
m.grad_buckets = adist.make_grad_buckets(m)
opt.grad_sync = m.grad_buckets.finish
random.seed("seed-ast-20h")
def one_step():
    with using_config("train", True):
        l = m.forward_loss(X=X, y=y, teach_ratio=0.8, random_out=0, add_noise=0.25)
        m.cleargrads(); l.backward(); opt.update()
    return l
l = one_step()
assert float(l.data) > 0 and float(m.grad_buckets.status_sum[0]) == 0.0            # healthy: the tail rides along and sums to zero
before = m.arena.data.clone()
m.grad_buckets.status_fn = lambda tail: tail.fill_(4.0)                            # "a peer's decoder forward timed out"
l = one_step()                                                                      # round 5: the MERGED word is written into THIS step's [loss, status] pair
torch.cuda.synchronize()
assert float(m.grad_buckets.status_sum[0]) == 4.0
assert torch.equal(before, m.arena.data), "the update of the aborted step was applied"
assert int(l.pair[1].item()) & 16, "the merged status word did not reach the pair of the step it belongs to"
m.grad_buckets.status_fn = adist._library_status
try:
    float(l.data)                                                                   # the read-back of the aborted step itself raises (on every rank alike)
    raise SystemExit("no AstkError after a peer's abort")
except _lib.AstkError as e:
    assert "peer rank" in str(e), str(e)
    print("abort propagated:", str(e)[:120])
assert torch.equal(before, m.arena.data), "updates were applied while the status word was set"
float(one_step().data)                                                              # the raise cleared the word: training can go on
assert not torch.equal(before, m.arena.data)
td.destroy_process_group()
# bench.py's own data-parallel branch (what the driver's N > 1 runs execute), forced onto a process group of one rank: RCCL backend,
# NCCL_MAX_NCHANNELS cap, bucketed exchange, max-over-ranks timing, the `dp` object of the JSON line
import json, subprocess
env = dict(os.environ, ASTK_BENCH_FORCE_DP="1", MASTER_PORT="29613")
env.pop("NCCL_MAX_NCHANNELS", None)
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline",
                    "--no-alt-precisions", "--profile-steps", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
print("bench dp branch:", line["dp"], line["ms_per_step"])
assert line["dp"]["rccl_ranks"] == 1 and line["dp"]["backend"] == "nccl" and set(line["dp"]["buckets"]) == {"cnn", "enc", "dec"}
assert line["dp"]["NCCL_MAX_NCHANNELS"] + line["dp"]["recurrence_grid_cus"] <= 256 and line["n_gpus"] == 1
print("ok")
