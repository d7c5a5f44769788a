"""NN drop-in (nn.py:42-322 of the reference): builds the model + optimizer from <cfg_dir>, resumes from the newest
seq2seq_<N>.model, and runs the train step of nn.py:168-194 -- forward_loss -> cleargrads -> backward -> update --
on the HIP path.  Data-parallel runs (one process per GPU, torchrun) shard every bucketed batch over the ranks and
all-reduce the flat gradient arena over RCCL before the hooks (ast_amd.dist)."""
import os
import random

import torch
from tqdm import tqdm

from . import dist as adist
from . import optimizers, serializers
from .config import Config
from .dataloader import SYMBOLS, FisherDataLoader, GlobalPhoneDataLoader, SyntheticDataLoader
from .seq2seq import SpeechEncoderDecoder, raise_if_aborted, using_config

_ADAM = 0
_SGD = 1


# ---- beam search (nn.py:235-322 of the reference): one utterance, N best hypotheses kept, K expansions per live hypothesis.
# Module-level so that it works on a bare SpeechEncoderDecoder too; NN.decode_beam / init_hyp / decode_beam_step delegate here.
def init_hyp(model):
    import torch
    return {"hyp": [SYMBOLS.GO_ID], "score": 0, "dec_state": model.get_encoder_states(),
            "attn_v": torch.zeros(1, model.cfg["rnn_config"]["attn_units"], dtype=torch.float32, device=model.device), "attn_history": []}


def decode_beam_step(model, decode_entry, beam_width):
    import torch
    with using_config("train", False):
        model.set_decoder_states(decode_entry["dec_state"])
        word = torch.full((1,), int(decode_entry["hyp"][-1]), dtype=torch.int32)
        logits, ht, alphas = model.decode_step(word, decode_entry["attn_v"])
        logp = torch.log_softmax(logits[0].double(), dim=0).cpu().numpy()
        top = logp.argsort()[-beam_width:]
        state = model.get_decoder_states()
        return [{"hyp": decode_entry["hyp"] + [int(pi)], "score": decode_entry["score"] + float(logp[pi]), "dec_state": state, "attn_v": ht,
                 "attn_history": decode_entry["attn_history"] + [alphas.squeeze().cpu().numpy()]} for pi in top[::-1]]


def decode_beam(model, X, stop_limit, N, K):
    with using_config("train", False):
        model.encode(X)
        n_best = [init_hyp(model)]
        for _ in range(stop_limit):
            if all(e["hyp"][-1] == SYMBOLS.EOS_ID for e in n_best):
                break
            cur = []
            for e in n_best:
                if e["hyp"][-1] != SYMBOLS.EOS_ID:
                    cur.extend(decode_beam_step(model, e, K))
                else:
                    cur.append(e)
            n_best = sorted(cur, reverse=True, key=lambda t: t["score"])[:N]
    return n_best


class NN:
    def __init__(self, cfg_path, vocab_size=None):
        self.cfg = Config(cfg_path, vocab_size=vocab_size)
        self.model_dir = self.cfg.model["model_dir"]
        self.gpuid = self.cfg.train["gpuid"]
        if adist.is_distributed():
            self.gpuid = adist.local_rank()           # one process per GPU: the rank picks the device
        random.seed(self.cfg.train["seed"])            # nn.py:54 -- the teacher-forcing / shuffling stream (Q4)
        data = self.cfg.train["data"]
        kind = data.get("dataloader", "fisher")
        loader = {"globalphone": GlobalPhoneDataLoader, "synthetic": SyntheticDataLoader}.get(kind, FisherDataLoader)
        self.data_loader = loader(data, self.model_dir, self.gpuid)
        if adist.is_distributed():
            self.data_loader.rank, self.data_loader.world = adist.rank(), adist.world_size()
        self.get_model()
        # extension key (BASELINE configs[4], "fp16 MFMA GEMMs"): extras.gemm_operands = "fp16" runs the batched products of the CNN
        # layers >= 1 and of the encoder's input projection with fp16 operands / f32 accumulation; default "f32" (f32-accurate products)
        # extras.gemm_precision = "bf16x3" (library default: exact f32 operands as three bf16 terms) | "f32" (f32-input MFMAs) | "fp16x2"
        # (two scaled fp16 terms: faster, narrower than float32).  Both go into the op descriptors of THIS model (no process-wide state).
        ops = self.cfg.train.get("extras", {}).get("gemm_operands", "f32")
        if ops not in ("f32", "fp16"):
            raise ValueError("extras.gemm_operands must be 'f32' or 'fp16'")
        prec = self.cfg.train.get("extras", {}).get("gemm_precision")
        if prec not in (None, "bf16x3", "f32", "fp16x2"):
            raise ValueError("extras.gemm_precision must be 'bf16x3', 'f32' or 'fp16x2'")
        self.model.gemm_operands, self.model.gemm_precision = ops, prec
        # extension key: extras.deterministic = true -> every gradient sum in a fixed order (include/astk.h `deterministic`; a few per cent slower)
        self.model.deterministic = bool(self.cfg.train.get("extras", {}).get("deterministic", False))
        self.init_optimizer(self.cfg.train["optimizer"])
        if self.cfg.train.get("save_optimizer", False) and self.loaded_from and self.model.arena is not None:
            # extension key: checkpoints also carry the Adam moments, so a resumed run continues instead of re-warming them
            if serializers.load_optimizer(self.loaded_from, self.model, self.optimizer):
                print("optimizer state restored (step {0:d})".format(self.optimizer.t))
        self.train_log = os.path.join(self.model_dir, "train.log")
        self.dev_log = os.path.join(self.model_dir, "dev.log")

    def init_optimizer(self, opt_cfg):
        print("Setting up optimizer")
        if opt_cfg["type"] == _ADAM:
            print("using ADAM")
            self.optimizer = optimizers.Adam(alpha=opt_cfg["lr"], beta1=0.9, beta2=0.999, eps=1e-08, amsgrad=True)
        else:
            print("using SGD")
            self.optimizer = optimizers.SGD(lr=opt_cfg["lr"])
        print("learning rate: {0:f}".format(opt_cfg["lr"]))
        self.optimizer.setup(self.model)
        if opt_cfg["l2"] > 0:
            print("Adding WeightDecay: {0:f}".format(opt_cfg["l2"]))
            self.optimizer.add_hook(optimizers.WeightDecay(opt_cfg["l2"]))
        print("Clipping gradients at: {0:d}".format(opt_cfg["grad_clip"]))
        self.optimizer.add_hook(optimizers.GradientClipping(threshold=opt_cfg["grad_clip"]))
        if opt_cfg["grad_noise_eta"] > 0:
            print("Adding gradient noise: {0:f}".format(opt_cfg["grad_noise_eta"]))
            self.optimizer.add_hook(optimizers.GradientNoise(eta=opt_cfg["grad_noise_eta"]))
        links = {n.split("/")[0] for n in (self.model.arena.shapes if self.model.arena is not None else [])}
        for l in opt_cfg["freeze"]:
            if not links or l in links:
                print("freezing: {0:s}".format(l))
                self.model[l].disable_update()
            else:
                print("layer {0:s} not in model".format(l))
        if adist.is_distributed():
            # replicas draw different dropout masks / speech noise for their different rows (the teacher-forcing stream stays common)
            self.model.rng_seed = (self.model.rng_seed + 0x9E3779B97F4A7C15 * adist.rank()) & 0xFFFFFFFFFFFFFFFF
            if self.model.arena is not None:
                self.model.grad_buckets = adist.make_grad_buckets(self.model)
                self.optimizer.grad_sync = self.model.grad_buckets.finish
            else:      # parameters materialise lazily on the first batch: fall back to one all-reduce of the whole arena
                self.optimizer.grad_sync = adist.allreduce_grads
            if self.cfg.train.get("sync_bn", False):    # extension key: BatchNorm statistics over the global batch
                self.model.stat_exchange = adist.StatExchange()

    def get_model(self):
        self.model_fname = os.path.join(self.model_dir, "seq2seq.model")
        self.model = SpeechEncoderDecoder(self.gpuid, self.cfg.model)
        self.model.to_gpu(self.gpuid)
        feat_dim = self.cfg.train["data"].get("feat_dim")
        if feat_dim:
            self.model.materialize(int(feat_dim), seed=0)     # same seed on every rank: replicas start identical
        self.max_epoch = 0
        self.loaded_from = None
        print("Checking for model in: {0:s}".format(self.model_dir))
        stem = os.path.basename(self.model_fname).replace(".model", "")
        files = [f for f in os.listdir(os.path.dirname(self.model_fname)) if stem in f and f.endswith(".model")]
        if files:
            newest = max(files, key=lambda s: int(s.split("_")[-1].split(".")[0]))
            path = os.path.join(os.path.dirname(self.model_fname), newest)
            print("model found = \n{0:s}".format(path))
            serializers.load_npz(path, self.model)
            self.loaded_from = path
            self.max_epoch = int(newest.split("_")[-1].split(".")[0])
        else:
            print("model not found")

    def train_epoch(self, set_key):
        total_loss, n_batches = 0.0, 0
        n_utts = self.data_loader.n_utts[set_key]
        ex = self.cfg.train["extras"]
        avg_loss = 0.0
        # nn.py:189 reads the loss back after every step (a device sync).  Here the read of step i happens after step i+1 has been
        # enqueued, so the device never waits for the host; the reported numbers are the same, the progress bar lags by one batch.
        pending = None

        def settle(p):
            nonlocal total_loss, n_batches, avg_loss
            vals = p[0].tolist()                                       # [loss, status word of the persistent kernels]
            raise_if_aborted(vals[1], "NN.train_epoch")
            loss_val = vals[0] / p[1]                                  # quirk Q5: divided by the batch size
            n_batches += 1
            total_loss += loss_val
            avg_loss = total_loss / n_batches
            pbar.set_description("loss={0:0.4f}".format(avg_loss))
            pbar.update(p[2] * self.data_loader.world)
        # (a stream of its own rather than the legacy default stream: slightly faster, and what lets the model's side stream run beside the recurrences)
        if getattr(self, "_compute_stream", None) is None:
            self._compute_stream = torch.cuda.Stream(device=self.model.device)
        torch.cuda.synchronize(self.model.device)
        with tqdm(total=n_utts, ncols=80, disable=adist.rank() != 0) as pbar, torch.cuda.stream(self._compute_stream):
            for batch in self.data_loader.get_batch(self.cfg.train["batch_size"], set_key, train=True, labels=True):
                with using_config("train", True):
                    loss = self.model.forward_loss(X=batch["X"], y=batch["y"], teach_ratio=ex["teach_ratio"],
                                                   random_out=ex["random_out"], add_noise=ex["speech_noise"],
                                                   y_global=batch.get("y_global"))
                    self.model.cleargrads()
                    loss.backward()
                    self.optimizer.update()
                cur = (loss.pair.clone(), len(batch["y"]), len(batch["X"]))   # the loss buffer is reused by the next step
                if pending is not None:
                    settle(pending)
                pending = cur
            if pending is not None:
                settle(pending)
        torch.cuda.synchronize(self.model.device)      # later default-stream work (predict, checkpoint) sees the epoch's updates
        return avg_loss

    # ---- nn.py:235-322
    def init_hyp(self):
        return init_hyp(self.model)

    def decode_beam_step(self, decode_entry, beam_width):
        return decode_beam_step(self.model, decode_entry, beam_width)

    def decode_beam(self, X, stop_limit, N, K):
        return decode_beam(self.model, X, stop_limit, N, K)

    def predict(self, set_key):
        preds = []
        stop_limit = self.cfg.train["data"]["max_pred"]
        with tqdm(total=self.data_loader.n_utts[set_key], ncols=80, disable=adist.rank() != 0) as pbar:
            for batch in self.data_loader.get_batch(self.cfg.train["batch_size"], set_key, train=False, labels=False):
                with using_config("train", False):
                    p = self.model.predict(batch["X"], SYMBOLS.GO_ID, SYMBOLS.EOS_ID, stop_limit)
                    preds.extend(zip(batch["utts"], p.tolist()))
                pbar.update(len(batch["X"]))
        return preds
