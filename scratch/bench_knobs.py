"""bench.py with tuning knobs from ASTK_BENCH_KNOBS="key=value,key=value" (SIDE=0 switches the model's side stream off): same-box A/B runs."""
import os, sys, runpy
import torch  # noqa: F401  (first: libastk.so must bind to the HIP runtime torch ships, not load /opt/rocm's beside it)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ast_amd import _lib
for kv in os.environ.get("ASTK_BENCH_KNOBS", "").split(","):
    if not kv:
        continue
    k, v = kv.split("=")
    if k == "SIDE":
        os.environ["ASTK_SIDE_STREAM"] = v
    else:
        _lib.set_tuning(k, float(v))
sys.argv[0] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
