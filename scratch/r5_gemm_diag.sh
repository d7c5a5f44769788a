#!/bin/bash
# in-kernel role stamps (debug build) and SQ counters of the product kernel on the step's big shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_diag; rm -rf $O; mkdir -p $O
SHAPES=("0 38400 512 1152" "0 6400 1024 3072" "1 6400 3072 1024" "2 1024 3072 6400 2" "0 4096 4096 4096")
for sh in "${SHAPES[@]}"; do
  echo "== $sh"
  python3 scratch/gemm_one.py $sh
  ASTK_LIB_PATH=$PWD/scratch/libastk_stamps.so python3 scratch/gemm_one.py $sh 2>&1 | tail -4
done
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 scratch/gemm_one.py $SH > $O/$n.log 2>&1; }
for i in 0 3; do
  SH=${SHAPES[$i]}
  run a$i SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  run b$i SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
  run c$i GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_LEVEL_LDS
done
python3 - <<PY
import csv,glob,collections
for d in ["a0","b0","c0","a3","b3","c3"]:
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_f32_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,len(v),round(sum(v)/len(v)))
    for f in glob.glob("$O/%s/**/*kernel_trace.csv"%d, recursive=True):
        ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "gemm_f32_kernel" in r["Kernel_Name"]]
        print(d,"mean kernel us",round(sum(ds)/len(ds),1))
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
