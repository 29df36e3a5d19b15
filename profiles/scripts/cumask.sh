#!/bin/bash
# experiment: can_see sweeps confined to a subset of the compute units (SW_CS_CUS)
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-cumask}; mkdir -p $out
B="--cpu-sample 0 --e2e-steps 0 --steps 10 --warmup 2"
for cus in 0 192 128 96; do
  SW_CS_CUS=$cus timeout 120 python bench.py $B > $out/bench_cus$cus.json 2> $out/err_$cus.log
  python - $out/bench_cus$cus.json $cus <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); k=d["roofline"]["kernels"]
    print("cus", sys.argv[2], "%.1f M ev/s %.3f ms |" % (d["value"]/1e6, d["ms_per_step"]), " ".join("%s %.1fus" % (x["kernel"][:16], x["avg_launch_us"]) for x in k[:3]))
except Exception as e: print("cus", sys.argv[2], "ERR", e)
PY
done
