#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-records}; mkdir -p $out
timeout 120 python profiles/exact_bench.py > $out/exact_bench.log 2>&1; cat $out/exact_bench.log | tail -4
B="--cpu-sample 0 --e2e-steps 0 --steps 5 --warmup 1"
timeout 100 python bench.py $B --members 1024 --events 2000000 > $out/bench_1024x2M.json 2> $out/err1.log
timeout 100 python bench.py $B --members 64 --events 100000 --steps 10 > $out/bench_64x100k.json 2> $out/err2.log
python - $out <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "%.1f M ev/s  %.3f ms" % (d["value"]/1e6, d["ms_per_step"]))
    except Exception as e: print(f, "ERR", e)
PY
