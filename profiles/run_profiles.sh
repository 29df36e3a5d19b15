#!/bin/bash
# Profile recipe of the bench workload (run on the GPU box from the repo root):
#   profiles/run_profiles.sh <tag> [bench args...]      (TRAFFIC_KEY=<members>x<events>x<mode> names the workload in traffic.json;
#   default 256x1000000x0 = bench.py's default; TRAFFIC_JSON=<file> extends that file instead of starting one under gpurun_out)
# writes gpurun_out/prof_<tag>/{kernel_stats.txt, pmc_summary.txt, traffic.json, loop_timeline.txt, *.log};
# copy what should be judged into profiles/<tag>_*.  Counter passes run on their own (no tracing
# domains besides the kernel trace), FETCH_SIZE and WRITE_SIZE in separate passes.
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--cpu-sample 0 --e2e-steps 0 $*"
PMCARGS="--steps 1 --warmup 0 --contexts 1 --cpu-sample 0 --e2e-steps 0 $*"
COMMIT=${SW_COMMIT:-unknown}
# 1. kernel trace of the bench command (default steps)
rocprofv3 --kernel-trace -d $OUT -o kt -- python bench.py $ARGS > $OUT/kt_run.log 2>&1
DB=$(ls $OUT/*kt_results.db $OUT/*/*kt_results.db 2>/dev/null | head -1)
python profiles/summarize_rocpd.py "$DB" > $OUT/kernel_stats.txt 2>> $OUT/kt_run.log
python profiles/loop_timeline.py "$DB" > $OUT/loop_timeline.txt 2>> $OUT/kt_run.log
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python bench.py $PMCARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python bench.py $PMCARGS > $OUT/write.log 2>&1
TJ=${TRAFFIC_JSON:-$OUT/traffic.json}
python profiles/collect_traffic.py --stats $OUT/kernel_stats.txt $OUT/fetch $OUT/write $TJ "$COMMIT" "${TRAFFIC_KEY:-256x1000000x0}" "bench.py $PMCARGS" \
    k_cansee_chunks k_cansee_fixup k_cansee_flow k_cansee_stream k_resolve_band k_tally_bits k_tally_tree k_elections k_voter_masks_bits k_finalize_events k_finalize_check k_finalize_listed \
    k_order_walk k_order_median k_order_sort k_order_bounds k_level_hist k_level_scatter k_level_patch > $OUT/traffic.log 2>&1
# 3. wave / wait / cache counters
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
    --output-format csv -d $OUT/pmc1 -o p1 -- python bench.py $PMCARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc2 -o p2 -- python bench.py $PMCARGS > $OUT/pmc2.log 2>&1
python profiles/summarize_pmc_csv.py $OUT/fetch $OUT/write $OUT/pmc1 $OUT/pmc2 > $OUT/pmc_summary.txt 2>&1
# keep the merge small: drop the raw databases / CSVs above a few MB
find $OUT -name '*.db' -size +8M -delete
find $OUT -name '*.csv' -size +8M -delete
echo done
