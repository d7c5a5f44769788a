#!/bin/bash
# Round-2 profile set (run on the GPU box through gpurun): kernel-trace statistics of the default bench command, three PMC passes
# (counters never combined with the trace domains gpurun refuses), the in-kernel phase stamps of the persistent decoder.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r2
rm -rf $OUT && mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/stats_es
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_es -- python3 bench.py --model es_en_20h --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_stats_es.log 2>&1
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
python3 scratch/pmc_summarize.py $OUT 4 > $OUT/pmc_summary.log 2>&1
ASTK_PERSIST_DBG=8 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > $OUT/phase_stamps.log 2>&1
ASTK_GEMM_PREC=bf16x3 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/bench_bf16x3.log 2>&1
ASTK_GEMM_PREC=f32 python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $OUT/bench_f32.log 2>&1
python3 bench.py --steps 50 --warmup 10 > $OUT/bench_default.log 2>&1
python3 bench.py --model es_en_20h --steps 50 --warmup 10 > $OUT/bench_es_en_20h.log 2>&1
# keep the summaries, drop the bulky per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls -R $OUT | head -50
