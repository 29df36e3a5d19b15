#!/bin/bash
# bench the 256 x 1M workload over an environment knob: sweep_cfg.sh <tag> <VAR> <values...>
O=gpurun_out/$1; mkdir -p $O; VAR=$2; shift 2
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
for v in "$@"; do
  env $VAR=$v timeout 200 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 10 --warmup 2 $BENCH_ARGS > $O/bench_${VAR}_$v.json 2>> $O/err.log
  python - $O/bench_${VAR}_$v.json "$VAR=$v" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); ks=d["roofline"]["kernels"]
    print(sys.argv[2], "%.1f M ev/s %.3f ms |" % (d["value"]/1e6, d["ms_per_step"]), " ".join("%s %.1fus x%d" % (k["kernel"][2:12], k["avg_launch_us"], k["launches"]) for k in ks[:3]), "| iters", d["roofline"]["counters"]["round_iterations"])
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
