#!/bin/bash
# GPU call r02b: dataflow can_see sweep (SW_CANSEE_IMPL=6, default) vs the level kernel (5)
O=gpurun_out/r02b; mkdir -p $O
(timeout 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
tail -14 $O/pytest_gpu.log
B="--cpu-sample 0 --e2e-steps 0 --steps 10 --warmup 2"
for impl in 6 5; do
  SW_CANSEE_IMPL=$impl timeout 200 python bench.py $B > $O/bench_256x1M_impl$impl.json 2> $O/err_$impl.log
  SW_CANSEE_IMPL=$impl timeout 200 python bench.py $B --members 1024 --events 2000000 --steps 5 > $O/bench_1024x2M_impl$impl.json 2>> $O/err_$impl.log
  SW_CANSEE_IMPL=$impl timeout 200 python bench.py $B --members 64 --events 100000 > $O/bench_64x100k_impl$impl.json 2>> $O/err_$impl.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02b/bench_*.json")):
    try:
        d=json.load(open(f)); k=d["roofline"]["kernels"][0]
        print(f.split("/")[-1], "%.1f M ev/s  %.3f ms | %s: %d launches avg %.1f us total %.2f ms" % (d["value"]/1e6, d["ms_per_step"], k["kernel"], k["launches"], k["avg_launch_us"], k["total_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
SW_DEBUG_TIMING=1 timeout 100 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 > /dev/null 2> $O/debug_timing.log; grep "sub-batch\|stages" $O/debug_timing.log | tail -8
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --cpu-sample 0 --e2e-steps 0 > $O/kt_run.log 2>&1
DB=$(ls $O/kt/*kt_results.db $O/kt/*/*kt_results.db 2>/dev/null | head -1)
python profiles/summarize_rocpd.py "$DB" > $O/kernel_stats.txt; head -8 $O/kernel_stats.txt
python profiles/loop_timeline.py "$DB" > $O/loop_timeline.txt; cat $O/loop_timeline.txt
find $O -name '*.db' -size +8M -delete
