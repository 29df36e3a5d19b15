#!/bin/bash
# round 3: 1024-member defaults (window of 12 slots from offset 10, wide elections) — parity and the bench lines
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_partition.py -m gpu -x -q > $O/pytest_1024.log 2>&1; echo "pytest rc=$?" >> $O/pytest_1024.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc|Error" $O/pytest_1024.log | cut -c1-300 | tail -6
B="--cpu-sample 0 --e2e-steps 0 --contexts 1"
timeout 200 python bench.py $B --steps 5 --warmup 2 --members 1024 --events 2000000 > $O/bench_1024x2M.json 2>> $O/err.log
timeout 900 python bench.py $B --steps 2 --warmup 0 --members 1024 --mode 2 --p0 0.40 --p1 0.02 --events 50000000 > $O/bench_c5_1024x50M.json 2>> $O/err.log
timeout 600 python bench.py $B --steps 2 --warmup 0 --members 1024 --mode 1 --p0 0.002 --events 20000000 > $O/bench_c5_1024x20M_two_cliques.json 2>> $O/err.log
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f)); c=d["config"]
        print(f.split("/")[-1], "%.1f M ev/s  %.3f ms | rounds %d coin votes %d" % (d["value"]/1e6, d["ms_per_step"], c["rounds"], c["coin_round_votes"]))
        for k in d["roofline"]["kernels"]: print("   ", k["kernel"], k["launches"], k["avg_launch_us"], k["total_ms"])
    except Exception as e: print(f, "ERR", e)
PY
