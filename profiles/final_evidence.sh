#!/bin/bash
# The evidence set of a round in ONE gpurun call (run from the repo root on the GPU box):
#   profiles/final_evidence.sh <tag>   -> gpurun_out/<tag>/...   (copy what should be judged into profiles/<tag>_*)
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export SW_COMMIT=${SW_COMMIT:-$(cat .commit_id 2>/dev/null || echo $TAG)}
# 1. the whole GPU suite and the smoke check
timeout 2400 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
# 2. profile recipe of the default workload and of 1024 members / 2 M events: kernel trace, PMC passes -> ONE traffic.json keyed by workload
TRAFFIC_JSON=$OUT/traffic.json TRAFFIC_KEY=256x1000000x0 profiles/run_profiles.sh ${TAG}_256 > /dev/null 2>&1
TRAFFIC_JSON=$OUT/traffic.json TRAFFIC_KEY=1024x2000000x0 profiles/run_profiles.sh ${TAG}_1024 --members 1024 --events 2000000 > /dev/null 2>&1
for w in 256 1024; do
    for f in kernel_stats.txt pmc_summary.txt loop_timeline.txt; do cp gpurun_out/prof_${TAG}_$w/$f $OUT/${w}_$f 2>/dev/null; done
done
cp $OUT/traffic.json profiles/traffic.json    # (bench.py below quotes it)
# 3. the bench lines
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --members 1024 --events 2000000 --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 --concurrent 0 --emulate-parts 2 > $OUT/bench_1024x2M_split2.json 2> $OUT/bench_1024x2M_split2.err
python bench.py --members 64 --events 100000 --cpu-sample 0 > $OUT/bench_64x100k.json 2> $OUT/bench_64x100k.err
python bench.py --members 1024 --events 2000000 --mode 2 --p0 0.4 --p1 0.02 --steps 3 --warmup 1 --cpu-sample 0 --e2e-steps 0 --concurrent 0 > $OUT/bench_1024x2M_coin_stress.json 2> $OUT/bench_1024x2M_coin_stress.err
# configs[4]'s size on ONE GPU (205 GB of can_see table: one context)
timeout 900 python bench.py --members 1024 --events 50000000 --mode 2 --p0 0.4 --p1 0.02 --steps 2 --warmup 0 --cpu-sample 0 --e2e-steps 0 --concurrent 0 --contexts 1 > $OUT/bench_c5_1024x50M.json 2> $OUT/bench_c5_1024x50M.err
# 4. find_order: laps, timeline, counters
ORDER_CALLS=12 python profiles/order_laps.py > $OUT/order_laps_256x1M.txt 2>&1
rocprofv3 --kernel-trace -d $OUT/okt -o kt -- python profiles/order_laps.py > $OUT/okt.log 2>&1
DB=$(ls $OUT/okt/*kt_results.db $OUT/okt/*/*kt_results.db 2>/dev/null | head -1)
python profiles/order_timeline.py "$DB" > $OUT/order_timeline_256x1M.txt 2>&1
python profiles/summarize_rocpd.py "$DB" | grep "k_order\|^kernel" > $OUT/order_kernel_stats_256x1M.txt 2>&1
SW_ORDER_ONE_STREAM=1 profiles/order_pmc.sh $TAG/opmc > /dev/null 2>&1
cp $OUT/opmc/order_pmc.txt $OUT/order_pmc_256x1M.txt 2>/dev/null
ORDER_CALLS=6 python profiles/order_laps.py 1024 2000000 > $OUT/order_laps_1024x2M.txt 2>&1
find $OUT gpurun_out/prof_${TAG}_256 gpurun_out/prof_${TAG}_1024 -name '*.db' -delete 2>/dev/null
find $OUT gpurun_out/prof_${TAG}_256 gpurun_out/prof_${TAG}_1024 -name '*.csv' -size +1M -delete 2>/dev/null
rm -rf $OUT/okt $OUT/opmc/pmc_*
echo done
