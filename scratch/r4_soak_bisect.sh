cd "$GRAFT_REPO_ROOT"
for v in "" "ASTK_CONV0_DIRECT=0" "ASTK_CNN_SEQ_BWD=0" "ASTK_CONV0_DIRECT=0 ASTK_CNN_SEQ_BWD=0"; do
  echo "== variant: $v"
  timeout -k 10 300 python3 scratch/soak.py cfg1 3000 $v 2>&1 | tail -n 2 | cut -c1-600
done
