#!/bin/bash
# Round-5 profile set (run on the GPU box through gpurun, ONE call, one box): kernel-trace statistics of the bench command under EVERY
# arithmetic scheme bench.py times (bf16x3 = the default and the headline, f32, fp16x2) and of the es_en_20h model, three PMC passes of
# the headline scheme (counters never combined with trace domains other than the kernel trace), the in-kernel phase stamps, then the
# bench lines BASELINE.md quotes.
set -e
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r5
rm -rf $OUT && mkdir -p $OUT
BS="--steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions --no-also"
for p in bf16x3 f32 fp16x2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$p -- python3 bench.py $BS --precision $p > $OUT/bench_stats_$p.log 2>&1
  echo "stats $p done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_es -- python3 bench.py --model es_en_20h $BS > $OUT/bench_stats_es.log 2>&1
echo "stats es done"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1
echo "pmc done"
python3 scratch/pmc_summarize.py $OUT 4 "$(git rev-parse --short HEAD 2>/dev/null || echo snapshot) $(date -u +%Y-%m-%dT%H:%MZ)" r5 bf16x3 > $OUT/pmc_summary.log 2>&1
# (round 5: the in-kernel phase timers exist in the TEST-HOOK build only -- the product library never reads ASTK_PERSIST_DBG)
ASTK_LIB_PATH=$PWD/ast_amd/libastk_test.so ASTK_PERSIST_DBG=8 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0 > $OUT/phase_stamps.log 2>&1
ASTK_LIB_PATH=$PWD/ast_amd/libastk_test.so ASTK_PERSIST_DBG=8 python3 bench.py --model es_en_20h --steps 2 --warmup 1 --no-cpu-baseline --no-alt-precisions --no-also --profile-steps 0 > $OUT/phase_stamps_es.log 2>&1
echo "stamps done"
python3 bench.py --steps 50 --warmup 10 > $OUT/bench_default.log 2>&1
echo "default done"
python3 bench.py --model cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_cfg5.log 2>&1
python3 bench.py --model cfg5 --gemm-operands fp16 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precisions > $OUT/bench_cfg5_fp16.log 2>&1
python3 bench.py --batch 64 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also > $OUT/bench_b64.log 2>&1
python3 bench.py --model cfg5 --hidden 2048 --steps 10 --warmup 3 --no-cpu-baseline --no-alt-precisions > $OUT/bench_cfg5_wide.log 2>&1
python3 bench.py --frames 1200 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also > $OUT/bench_t1200.log 2>&1
python3 bench.py --frames 1680 --steps 30 --warmup 10 --no-cpu-baseline --no-alt-precisions --no-also > $OUT/bench_t1680.log 2>&1
echo "matrix done"
PROF_ARGS="--no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt $OUT/trace_step_cfg1.txt
PROF_ARGS="--model es_en_20h --no-also --no-alt-precisions" bash scratch/trace_step.sh && cp gpurun_out/trace_step.txt $OUT/trace_step_es_en_20h.txt
bash scratch/gemm_step_table.sh > $OUT/gemm_step_table.txt 2>&1
ASTK_GEMM_HYBRID=0 ASTK_GEMM_TILE=128 bash scratch/gemm_step_table.sh > $OUT/gemm_step_table_r4_schedule.txt 2>&1
echo "traces done"
# keep the summaries, drop the bulky per-dispatch traces
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls -R $OUT | head -80
