#!/bin/bash
# round 4, last evidence at HEAD: whole GPU suite, default bench, profile recipe (traffic.json for this kernel source), find_order on the shapes with large rounds
O=gpurun_out/r04_final4; mkdir -p $O; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu.log | cut -c1-300 | tail -8
SW_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) timeout 600 bash profiles/run_profiles.sh r04_final4 > $O/prof.log 2>&1
cp gpurun_out/prof_r04_final4/traffic.json profiles/traffic.json   # (so that the bench lines below carry PMC bytes of this code)
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
B="--cpu-sample 0 --e2e-steps 0 --contexts 1 --concurrent 0"
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 1 --p0 0.02 > $O/bench_two_cliques_256x1M.json 2>> $O/err.log
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 2 --p0 0.35 --p1 0.02 > $O/bench_coin_stress_256x1M.json 2>> $O/err.log
timeout 300 python bench.py $B --steps 3 --warmup 1 --mode 2 --p0 0.95 --p1 0.002 > $O/bench_hot_members_256x1M.json 2>> $O/err.log
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        s=open(f).read(); d=json.loads(s[s.index('{"metric"'):]); r=d["roofline"]
        print(f.split("/")[-1], "%.1f M ev/s  %.3f ms | find_order %.2f ms (first %.2f) with order %.1f M | stale %s pmc %s" % (d["value"]/1e6, d["ms_per_step"], d["find_order_ms"], d["find_order_first_call_ms"], (d.get("value_with_order") or 0)/1e6, r["traffic_stale"], r["hbm_bytes_per_step_pmc"]))
    except Exception as e: print(f, "ERR", e)
PY
