"""Config drop-in (config.py:15-29 of the reference): <cfg_dir>/model_cfg.json + train_cfg.json, with the
decoder vocabulary size injected from the vocab pickle and `model_dir` set to the experiment directory."""
import json
import os
import pickle


class Config:
    def __init__(self, cfg_path, vocab_size=None):
        with open(os.path.join(cfg_path, "model_cfg.json"), "r") as f:
            self.model = json.load(f)
        with open(os.path.join(cfg_path, "train_cfg.json"), "r") as f:
            self.train = json.load(f)
        data = self.train.get("data", {})
        if vocab_size is None:
            if data.get("dataloader") == "synthetic":
                vocab_size = int(data["vocab_size"])
            else:
                with open(data["vocab_path"], "rb") as f:
                    vocab = pickle.load(f)
                vocab_size = len(vocab[data["dec_key"]]["w2i"])
        self.model["rnn_config"]["dec_vocab_size"] = vocab_size
        print("vocab size {0:s} = {1:d}".format(str(data.get("dec_key", "synthetic")), vocab_size))
        self.model["model_dir"] = cfg_path
