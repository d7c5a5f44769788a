#!/bin/bash
# Round-6 profile set (run on the GPU box through gpurun, ONE call, one box): kernel-trace statistics of the bench command (side stream OFF:
# every kernel alone on the device, the state `roofline` is defined on; and ON: the schedule the headline runs), under every arithmetic
# scheme and for the es_en_20h model; PMC passes of the headline scheme (counters never combined with trace domains other than the kernel
# trace); in-kernel phase stamps (instrumented build); the bench lines BASELINE.md quotes; ordered step traces with start offsets and queue
# ids (the overlapping launches of the side stream); the per-launch GEMM table; same-box A/B of the side-stream schedule.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r6
rm -rf $OUT && mkdir -p $OUT
BS="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions --no-also --histogram none"
export ASTK_SIDE_STREAM=0
for p in bf16x3 f32 fp16x2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$p -- python3 bench.py $BS --precision $p > $OUT/bench_stats_$p.log 2>&1
  echo "stats $p done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_es -- python3 bench.py --model es_en_20h $BS > $OUT/bench_stats_es.log 2>&1
echo "stats es done"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0 --histogram none"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
echo "pmc done"
python3 scratch/pmc_summarize.py $OUT 4 "$(git rev-parse --short HEAD 2>/dev/null || echo snapshot) $(date -u +%Y-%m-%dT%H:%MZ)" r6 bf16x3 > $OUT/pmc_summary.log 2>&1
unset ASTK_SIDE_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_side -- python3 bench.py $BS > $OUT/bench_stats_side.log 2>&1
echo "stats side-stream done"
# (the in-kernel phase timers exist in the TEST-HOOK build only -- the product library never reads ASTK_PERSIST_DBG)
ASTK_SIDE_STREAM=0 ASTK_LIB_PATH=$PWD/ast_amd/libastk_test.so ASTK_PERSIST_DBG=8 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0 --histogram none > $OUT/phase_stamps.log 2>&1
ASTK_SIDE_STREAM=0 ASTK_LIB_PATH=$PWD/ast_amd/libastk_test.so ASTK_PERSIST_DBG=8 python3 bench.py --model es_en_20h --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0 --histogram none > $OUT/phase_stamps_es.log 2>&1
echo "stamps done"
python3 bench.py --steps 50 --warmup 10 > $OUT/bench_default.log 2>&1
echo "default done"
ASTK_SIDE_STREAM=0 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-precisions --histogram none > $OUT/bench_inline.log 2>&1
ASTK_BENCH_KNOBS="gemm.deterministic=1,SIDE=0" python3 scratch/bench_knobs.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --histogram none > $OUT/bench_deterministic.log 2>&1
python3 bench.py --bucket-batch 64,48,32 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_bucket_batch.log 2>&1
python3 bench.py --model cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions --histogram none > $OUT/bench_cfg5.log 2>&1
python3 bench.py --model cfg5 --gemm-operands fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions --histogram none > $OUT/bench_cfg5_fp16.log 2>&1
python3 bench.py --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_b64.log 2>&1
ASTK_BENCH_KNOBS="lstm.rows32=0" python3 scratch/bench_knobs.py --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_b64_rows16.log 2>&1
ASTK_BENCH_KNOBS="lstm.rows32=1" python3 scratch/bench_knobs.py --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_b64_mt2.log 2>&1
ASTK_BENCH_KNOBS="lstm.rows32=2,SIDE=0" python3 scratch/bench_knobs.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_duo_inline.log 2>&1
ASTK_BENCH_KNOBS="lstm.rows32=1,SIDE=0" python3 scratch/bench_knobs.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_mt2_inline.log 2>&1
ASTK_BENCH_KNOBS="lstm.rows32=2,lstm.side_bwd=1" python3 scratch/bench_knobs.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_duo_side.log 2>&1
python3 bench.py --model cfg5 --hidden 2048 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions --histogram none > $OUT/bench_cfg5_wide.log 2>&1
python3 bench.py --frames 1200 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_t1200.log 2>&1
python3 bench.py --frames 1680 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also --histogram none > $OUT/bench_t1680.log 2>&1
echo "matrix done"
PROF_ARGS="" bash scratch/trace2.sh prof_r6/trace_step_cfg1
PROF_ARGS="--model es_en_20h" bash scratch/trace2.sh prof_r6/trace_step_es_en_20h
ASTK_SIDE_STREAM=0 PROF_ARGS="" bash scratch/trace2.sh prof_r6/trace_step_cfg1_inline
bash scratch/gemm_step_table.sh > $OUT/gemm_step_table.txt 2>&1
echo "traces done"
bash scratch/r6_ab_knobs.sh prof_r6/ab_side "inline:SIDE=0" "dec_only:lstm.side_fwd=0" "side:SIDE=1" > /dev/null 2>&1 || true
echo "ab done"
# keep the summaries, drop the bulky per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls -R $OUT | head -100
