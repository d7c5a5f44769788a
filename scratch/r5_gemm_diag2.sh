#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_diag2; rm -rf $O; mkdir -p $O
SH="0 4096 4096 4096 0 40"
for t in 128 256; do
  echo "== tile $t stamps"; ASTK_GEMM_TILE=$t ASTK_LIB_PATH=$PWD/scratch/libastk_stamps.so python3 scratch/gemm_one.py $SH 2>&1 | tail -3
  ASTK_GEMM_TILE=$t rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/p$t -- python3 scratch/gemm_one.py $SH > $O/p$t.log 2>&1
  ASTK_GEMM_TILE=$t rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/q$t -- python3 scratch/gemm_one.py $SH > $O/q$t.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in ["p128","q128","p256","q256"]:
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_f32_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,len(v),round(sum(v)/len(v)))
    for f in glob.glob("$O/%s/**/*kernel_trace.csv"%d, recursive=True):
        ds=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "gemm_f32_kernel" in r["Kernel_Name"]]
        print(d,"mean kernel us",round(sum(ds)/len(ds),1), "last", ds[-3:])
PY
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
