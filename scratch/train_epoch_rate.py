"""Frames/s of NN.train_epoch on the benchmark model when every batch starts in HOST memory (synthetic corpus in the reference's
schemas, bucket 9 lengths 720-879 frames): padding, pinned staging and the H2D copy are inside the measurement."""
import copy, json, os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from ast_amd.nn import NN
d = tempfile.mkdtemp()
mcfg = copy.deepcopy(bench.MODEL_CFG); mcfg["rnn_config"].pop("dec_vocab_size", None)
tcfg = {"seed": "seed-ast-20h", "gpuid": 0, "batch_size": 32, "train_set": "syn_train", "dev_set": "syn_dev", "iters_save": 1,
        "optimizer": {"type": 0, "lr": 1e-3, "l2": 1e-4, "grad_clip": 2, "grad_noise_eta": 0, "freeze": []},
        "extras": {"teach_ratio": 0.8, "random_out": 0, "speech_noise": 0.25},
        "data": {"dataloader": "synthetic", "vocab_size": 1098, "feat_dim": 80, "n_utts": {"syn_train": 1280, "syn_dev": 4},
                 "frames": [720, 799], "targets": [20, 38], "buckets_num": 20, "buckets_width": 80, "max_pred": 40,
                 "zero_input": 0.1, "train_scale": 1, "dec_key": "bpe_w"}}
json.dump(mcfg, open(d + "/model_cfg.json", "w")); json.dump(tcfg, open(d + "/train_cfg.json", "w"))
nn = NN(d)
frames = sum(v["sp"] for v in nn.data_loader.info["syn_train"].values())
nn.train_epoch("syn_train")                       # warm-up epoch (workspaces, first-touch)
torch.cuda.synchronize(); t0 = time.perf_counter()
loss = nn.train_epoch("syn_train")
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"\nepoch of 40 batches x 32 utterances: {dt*1e3/40:.2f} ms per batch, {frames/dt/1e6:.2f} M real frames/s ({40*32*800/dt/1e6:.2f} M padded-to-800 frames/s), loss {loss:.3f}")
