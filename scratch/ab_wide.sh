#!/bin/bash
cd "$GRAFT_REPO_ROOT"
ms() { python3 -c "import json,sys; print(json.loads([l for l in sys.stdin if l.startswith('{')][-1])['ms_per_step'])"; }
for i in 1 2; do
  echo -n "wide "; python3 bench.py --model cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions --profile-steps 0 2>/dev/null | ms
  echo -n "ASTK_DEC_WIDE=0 "; ASTK_DEC_WIDE=0 python3 bench.py --model cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions --profile-steps 0 2>/dev/null | ms
done
