#!/bin/bash
# round 3, first GPU call: the chunk-parallel can_see sweep — its own tests, the parity subset, bench lines
# chunked (default) against unchunked (SW_CHUNKS=1), sub-batch timing of both
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 500 python -m pytest tests/test_gpu_chunks.py -m gpu -q > $O/pytest_chunks.log 2>&1; echo "pytest rc=$?" >> $O/pytest_chunks.log)
grep -E "^(FAILED|ERROR|E  +Assertion|[0-9]+ (passed|failed))|AssertionError: can_see" $O/pytest_chunks.log | cut -c1-400 | tail -30
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.log)
tail -4 $O/pytest_subset.log
B="--cpu-sample 0 --e2e-steps 0 --steps 10 --warmup 2"
for cfg in "4 0" "1 0" "4 1" "4 2" "2 1" "8 0"; do
  set -- $cfg
  SW_CHUNKS=$1 SW_CHUNK_CFG=$2 timeout 200 python bench.py $B > $O/bench_256x1M_chunks$1_cfg$2.json 2>> $O/err.log
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f)); k=d["roofline"]["kernels"][0]
        print(f.split("/")[-1], "%.1f M ev/s  %.3f ms | %s: %d launches avg %.1f us total %.2f ms" % (d["value"]/1e6, d["ms_per_step"], k["kernel"], k["launches"], k["avg_launch_us"], k["total_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
for ch in 4 1; do
  SW_CHUNKS=$ch SW_DEBUG_TIMING=1 timeout 100 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 > /dev/null 2> $O/debug_timing_chunks$ch.log; echo "chunks=$ch"; grep "sub-batch\|stages" $O/debug_timing_chunks$ch.log | tail -6
done
