"""Chainer-compatible checkpoints (train.py:73-75, nn.py:141-152): compressed .npz with '<link>/<param>' keys,
interleaved-gate LSTM layout, BN persistents avg_mean / avg_var / N."""
import numpy as np


def save_npz(path, model, optimizer=None):
    out = {k: v.detach().cpu().numpy() for k, v in model.arena.views.items()}
    for k, v in model.persist.items():
        out[k] = v.detach().cpu().numpy()
    for i in range(len(model.cnns) if model.cnn_bn else 0):
        # Chainer's BatchNormalization counts its training-mode calls in the persistent N (A10); every layer sees every train forward
        out[f"CNN_{i}_bn/N"] = np.asarray(int(model.bn_N), dtype=np.int64)
    if getattr(model, "rnn_linear_proj", False):
        for i, n in enumerate(model.proj_bn_N):              # the projection's BatchNormalization is called once per TIME STEP
            out[f"enc_proj{i}_bn/N"] = np.asarray(int(n), dtype=np.int64)
    if optimizer is not None and getattr(optimizer, "m", None) is not None:
        # an extension over the reference (which never saves Adam state); Chainer's load_npz ignores extra keys
        out["__opt__/t"] = np.asarray(optimizer.t)
        # where the dropout / speech-noise stream stands: a resumed run must not replay the masks of its first epochs
        out["__opt__/rng_seed"] = np.asarray(int(model.rng_seed), dtype=np.uint64)
        out["__opt__/rng_offset"] = np.asarray(int(model._rng_offset), dtype=np.uint64)
        out["__opt__/m"] = optimizer.m.cpu().numpy()
        out["__opt__/v"] = optimizer.v.cpu().numpy()
        out["__opt__/vhat"] = optimizer.vhat.cpu().numpy()
    with open(path, "wb") as f:
        np.savez_compressed(f, **out)


def infer_in_dim(cfg, l0_fan_in):
    """Input feature dim from the first LSTM's fan-in C_last*F' (the reference's links are lazily shaped)."""
    cc = cfg["cnn_config"]["cnn_layers"]
    f2 = l0_fan_in // cc[-1]["out_channels"]
    kw, sw = cc[0]["ksize"][1], cc[0]["stride"][1]
    return (f2 - 1) * sw + kw


def load_npz(path, model, optimizer=None):
    import torch
    with np.load(path) as z:
        keys = set(z.files)
        if model.arena is None:
            model.V = int(z["out/W"].shape[0])
            model.cfg["rnn_config"]["dec_vocab_size"] = model.V
            model.materialize(infer_in_dim(model.cfg, z["L0_enc/upward/W"].shape[1]))
        for k, v in model.arena.views.items():
            if k not in keys:
                raise KeyError(f"{path}: missing parameter {k}")
            v.copy_(torch.from_numpy(np.asarray(z[k], dtype=np.float32)).reshape(v.shape))
        for k, v in model.persist.items():
            if k in keys:
                v.copy_(torch.from_numpy(np.asarray(z[k], dtype=np.float32)))
        if "CNN_0_bn/N" in keys:
            model.bn_N = int(z["CNN_0_bn/N"])
        for i in range(len(getattr(model, "proj_bn_N", []))):
            if f"enc_proj{i}_bn/N" in keys:
                model.proj_bn_N[i] = int(z[f"enc_proj{i}_bn/N"])
        if optimizer is not None:
            _restore_optimizer(z, keys, model, optimizer)


def _restore_optimizer(z, keys, model, optimizer):
    import torch
    if "__opt__/m" not in keys:
        return False
    optimizer.t = int(z["__opt__/t"])
    dev = model.arena.device
    optimizer.m = torch.from_numpy(z["__opt__/m"]).to(dev)
    optimizer.v = torch.from_numpy(z["__opt__/v"]).to(dev)
    optimizer.vhat = torch.from_numpy(z["__opt__/vhat"]).to(dev)
    if "__opt__/rng_offset" in keys:
        model.rng_seed, model._rng_offset = int(z["__opt__/rng_seed"]), int(z["__opt__/rng_offset"])
    return True


def load_optimizer(path, model, optimizer):
    """Restores the Adam/AMSGrad moments and step count saved by save_npz(..., optimizer=) -- an extension: the reference's
    checkpoints hold the model only, so a resumed run restarts its moments (train.py:73-75).  Returns False if the file has none."""
    with np.load(path) as z:
        return _restore_optimizer(z, set(z.files), model, optimizer)
