"""train.py drop-in (train.py:17-76 of the reference): python train.py -m <cfg_dir> -e <epochs>."""
import argparse
import os

from ast_amd import dist as adist
from ast_amd import serializers
from ast_amd.nn import NN

if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Train and evaluate NN model")
    parser.add_argument("-m", "--cfg_path", help="path for model config", required=True)
    parser.add_argument("-e", "--epochs", help="num epochs", required=True)
    args = vars(parser.parse_args())
    cfg_path, epochs = args["cfg_path"], int(args["epochs"])
    print("number of epochs={0:d}".format(epochs))
    adist.init()
    nn = NN(cfg_path)
    train_key, dev_key = nn.cfg.train["train_set"], nn.cfg.train["dev_set"]
    iters_save = nn.cfg.train["iters_save"]
    metrics = None
    refs_path = os.path.join(nn.cfg.train["data"].get("refs_path", ""), dev_key)
    if os.path.exists(os.path.join(refs_path, "eval.ids")):
        try:
            from ast_amd.eval import Eval
            metrics = Eval(refs_path, nn.cfg.train["data"]["n_evals"])
        except ImportError as e:                     # nltk is an optional dependency (BLEU is outside the hot path)
            print("BLEU disabled:", e)
    start_epoch = nn.max_epoch + 1
    max_epoch = start_epoch + epochs
    for epoch in range(start_epoch, max_epoch):
        print("-" * 80)
        print("Experment: {0:s} epoch: {1:d} gpu: {2:d}".format(cfg_path, epoch, nn.gpuid))
        print("-" * 80)
        # OLD-path extra (nmt_run.py:846-853): from epoch `iter_weight_noise` on, Gaussian noise on the LSTM / embedding weights before
        # every epoch (keys iter_weight_noise, weight_noise_mean, weight_noise_sigma in train_cfg.json; absent or 0: off)
        t_cfg = nn.cfg.train
        if t_cfg.get("iter_weight_noise", 0) > 0 and epoch >= t_cfg["iter_weight_noise"]:
            print("Adding Gaussian weight noise, mean={0:.2f}, stdev={1:0.6f}".format(t_cfg.get("weight_noise_mean", 0.0), t_cfg["weight_noise_sigma"]))
            if nn.model.arena is not None:
                nn.model.add_weight_noise(t_cfg.get("weight_noise_mean", 0.0), t_cfg["weight_noise_sigma"])
                if adist.is_distributed():
                    adist.broadcast_params(nn.model.arena)      # one draw for all replicas
        epoch_loss = nn.train_epoch(train_key)
        if adist.rank() == 0:
            with open(nn.train_log, mode="a") as f:
                f.write("{0:d}, {1:.4f}\n".format(epoch, epoch_loss))
        if metrics is not None and adist.rank() == 0:
            hyps = nn.data_loader.get_hyps(nn.predict(dev_key))
            bleu = metrics.calc_bleu(hyps) * 100
            with open(nn.dev_log, mode="a") as f:
                f.write("{0:d}, {1:.2f}\n".format(epoch, bleu))
            print("BLEU = {0:.2f}".format(bleu))
        if (epoch % iters_save == 0 or epoch == max_epoch - 1) and adist.rank() == 0:
            print("Saving model")
            serializers.save_npz(nn.model_fname.replace(".model", "_{0:d}.model".format(epoch)), nn.model,
                                 optimizer=nn.optimizer if nn.cfg.train.get("save_optimizer", False) else None)
            print("Finished saving model")
        # data parallel: the dev pass and the checkpoint are rank 0's alone; nobody enters the next epoch's all-reduces before it is back
        adist.barrier()
