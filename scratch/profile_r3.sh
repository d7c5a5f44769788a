#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun): kernel-trace statistics of the default bench command (and of the es_en_20h
# model), three PMC passes (counters never combined with the trace domains gpurun refuses), the in-kernel phase stamps of the
# persistent kernels, then the bench lines of every workload BASELINE.md quotes -- all from ONE box, one call.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r3
rm -rf $OUT && mkdir -p $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/stats_es
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_es -- python3 bench.py --model es_en_20h --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_stats_es.log 2>&1
echo "stats es done"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
echo "pmc done"
python3 scratch/pmc_summarize.py $OUT 4 "$(git rev-parse --short HEAD 2>/dev/null || echo snapshot) $(date -u +%Y-%m-%dT%H:%MZ)" > $OUT/pmc_summary.log 2>&1
ASTK_PERSIST_DBG=8 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --profile-steps 0 > $OUT/phase_stamps.log 2>&1
echo "stamps done"
python3 bench.py --steps 50 --warmup 10 > $OUT/bench_default.log 2>&1
echo "default done"
python3 bench.py --model es_en_20h --steps 50 --warmup 10 > $OUT/bench_es_en_20h.log 2>&1
echo "es_en_20h done"
python3 bench.py --model cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_cfg5.log 2>&1
python3 bench.py --model cfg5 --gemm-operands fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_cfg5_fp16.log 2>&1
python3 bench.py --gemm-operands fp16 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions > $OUT/bench_cfg1_fp16.log 2>&1
python3 bench.py --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions > $OUT/bench_b64.log 2>&1
python3 bench.py --frames 1200 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions > $OUT/bench_t1200.log 2>&1
python3 bench.py --frames 1680 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions > $OUT/bench_t1680.log 2>&1
python3 bench.py --model es_en_20h --frames 1200 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions > $OUT/bench_es_t1200.log 2>&1
echo "matrix done"
# keep the summaries, drop the bulky per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls -R $OUT | head -60
