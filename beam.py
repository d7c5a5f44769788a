"""beam.py drop-in (beam.py:46-146 of the reference):
    python beam.py -m <cfg_dir> -n <N hyps kept> -k <K candidates per step> -s <set key> -w <length weight> [--resume]
Beam search over one set with the newest checkpoint of the experiment, n-best lists pickled to
<cfg_dir>/<set>_beam_N-<N>_K-<K>.p, the length-normalised best hypothesis of each utterance scored with corpus BLEU
(ast_amd.eval) and written to <cfg_dir>/<set>_beam_N-<N>_K-<K>_W-<W>.en."""
import argparse
import math
import os
import pickle
import random

from tqdm import tqdm

from ast_amd.eval import Eval
from ast_amd.nn import NN


def rerank_hypothesis(beam_hyps, weight):
    """score / (len - 2)^weight, best first (beam.py:32-34).  A hypothesis of just [GO, EOS] has len - 2 = 0, which divides by
    zero in the reference; it is ranked with a length of 1 here."""
    return sorted([(h[0], h[1] / math.pow(max(len(h[0]) - 2, 1), weight), len(h[0])) for h in beam_hyps], reverse=True, key=lambda t: t[1])


def get_best_hyps(utts_beam, W):
    return {u: list(rerank_hypothesis(hyps, weight=W)[0][0]) for u, hyps in utts_beam.items()}


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Beam search to find best predictions for NN model")
    parser.add_argument("-m", "--cfg_path", help="path for model config", required=True)
    parser.add_argument("-n", "--N", help="number of hyps", required=True)
    parser.add_argument("-k", "--K", help="softmax selection", required=True)
    parser.add_argument("-s", "--S", help="dev/dev2/test", required=True)
    parser.add_argument("-w", "--W", help="len normalization weight", required=True)
    parser.add_argument("--resume", action="store_true", help="re-score the saved beam results instead of decoding again")
    args = vars(parser.parse_args())
    cfg_path, N, K, W, set_key = args["cfg_path"], int(args["N"]), int(args["K"]), float(args["W"]), args["S"]
    nn = NN(cfg_path)
    metrics = Eval(os.path.join(nn.cfg.train["data"]["refs_path"], set_key), nn.cfg.train["data"]["n_evals"])
    random.seed("meh")
    print("-" * 80)
    print("Beam for: {0:s} gpu: {1:d}".format(cfg_path, nn.gpuid))
    print("-" * 80)
    beam_fname = os.path.join(cfg_path, "{0:s}_beam_N-{1:d}_K-{2:d}.p".format(set_key, N, K))
    if args["resume"]:
        print("Loading saved beam results")
        with open(beam_fname, "rb") as f:
            beam = pickle.load(f)
    else:
        print("Computing beam results")
        stop_limit = nn.cfg.train["data"]["max_pred"]
        beam = {}
        with tqdm(total=nn.data_loader.n_utts[set_key], ncols=80) as pbar:
            for utt in nn.data_loader.get_batch(1, set_key, train=False, labels=False):
                n_best = nn.decode_beam(utt["X"], stop_limit=stop_limit, N=N, K=K)
                beam[utt["utts"][0]] = [(e["hyp"], e["score"], e["attn_history"]) for e in n_best]
                pbar.update(len(utt["X"]))
        print("saving hyps")
        with open(beam_fname, "wb") as f:
            pickle.dump(beam, f)
    hyps = nn.data_loader.get_hyps(get_best_hyps(beam, W).items())
    print("BLEU = {0:.2f}".format(metrics.calc_bleu(hyps) * 100))
    out_fname = os.path.join(cfg_path, "{0:s}_beam_N-{1:d}_K-{2:d}_W-{3:.2f}.en".format(set_key, N, K, W))
    metrics.write_to_file(hyps, out_fname)
    print("Predictions written to: {0:s}".format(out_fname))
