"""Length bucketing (preprocessing/prep_buckets.py:41-108): bucket = min(frames // width, num-1); optional
down-sampling of train sets with Python's seeded `random`; result pickled to <model_dir>/buckets_<key>.dict."""
import os
import pickle
import random


def create_buckets(cat_dict, num_b, width_b, key, scale=1, seed="haha"):
    buckets = [[] for _ in range(num_b)]
    for utt, info in cat_dict.items():
        buckets[min(info[key] // width_b, num_b - 1)].append(utt)
    if scale > 1:
        random.seed(seed)
        buckets = [random.sample(b, int(len(b) // scale)) for b in buckets]
    return {"buckets": buckets, "num_b": num_b, "width_b": width_b}


def buckets_from_info(info_dict, num_b, width_b, key="sp", scale=1, seed="haha"):
    return {cat: create_buckets(d, num_b, width_b, key, scale if "train" in cat else 1, seed) for cat, d in info_dict.items()}


def buckets_main(save_path, num_b, width_b, key, scale=1, seed="haha", info_path=""):
    if not os.path.exists(save_path):
        print("{0:s} does not exist. Exiting".format(save_path))
        return 0
    if not os.path.exists(info_path):
        print("{0:s} does not exist. Exiting".format(info_path))
        return 0
    with open(info_path, "rb") as f:
        info_dict = pickle.load(f)
    out = buckets_from_info(info_dict, num_b, width_b, key, scale, seed)
    with open(os.path.join(save_path, "buckets_{0:s}.dict".format(key)), "wb") as f:
        pickle.dump(out, f)
    return out
