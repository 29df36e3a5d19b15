#!/bin/bash
# quick GPU check: parity subset + bench lines of the can_see variants (args: tag)
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
if [ -z "$SKIP_TESTS" ]; then (timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log); fi
tail -4 $O/pytest_subset.log
B="--cpu-sample 0 --e2e-steps 0 --steps 10 --warmup 2"
timeout 200 python bench.py $B > $O/bench_256x1M.json 2> $O/err.log
timeout 200 python bench.py $B --members 1024 --events 2000000 --steps 5 > $O/bench_1024x2M.json 2>> $O/err.log
timeout 200 python bench.py $B --members 64 --events 100000 > $O/bench_64x100k.json 2>> $O/err.log
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f)); k=d["roofline"]["kernels"][0]
        print(f.split("/")[-1], "%.1f M ev/s  %.3f ms | %s: %d launches avg %.1f us total %.2f ms" % (d["value"]/1e6, d["ms_per_step"], k["kernel"], k["launches"], k["avg_launch_us"], k["total_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
SW_DEBUG_TIMING=1 timeout 100 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 > /dev/null 2> $O/debug_timing.log; grep "sub-batch\|stages" $O/debug_timing.log | tail -6
