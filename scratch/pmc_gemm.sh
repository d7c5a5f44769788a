#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcg; rm -rf $O; mkdir -p $O
run() { n=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 scratch/gemm_bench_one.py > $O/$n.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES
run c SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run d TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum
run e GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv,glob,collections
for d in "abcde":
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm_f32_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d,k,len(v),round(sum(v)/len(v)))
PY
