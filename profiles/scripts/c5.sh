#!/bin/bash
# BASELINE.json configs[4] shape on ONE GPU: 1024 members, coin-round stress (35 % of the members 50x less
# active), as many events as fit; 256-member / 10 M line for configs[3]
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
C5="--members 1024 --mode 2 --p0 0.35 --p1 0.02 --contexts 1 --cpu-sample 0 --e2e-steps 0 --warmup 0"
for N in 4000000 20000000 50000000; do
  /usr/bin/time -v timeout 600 python bench.py $C5 --events $N --steps 2 > $O/bench_c5_1024x$N.json 2> $O/c5_$N.err
  python - $O/bench_c5_1024x$N.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(c["members"], c["events"], "%.1f M ev/s %.1f ms | rounds %d coin votes %d (flips %d) | ingest %.2f s" % (d["value"]/1e6, d["ms_per_step"], c["rounds"], c["coin_round_votes"], c["coin_round_votes_from_signature_bit"], c["ingest_s_untimed"]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
  grep -i "Maximum resident\|error\|Error" $O/c5_$N.err | head -3
done
timeout 300 python bench.py --members 256 --events 10000000 --contexts 1 --cpu-sample 0 --e2e-steps 1 --warmup 1 --steps 3 > $O/bench_c4_256x10M.json 2> $O/c4.err
python -c "
import json; d=json.load(open('$O/bench_c4_256x10M.json')); print('256 x 10M: %.1f M ev/s %.1f ms, e2e %.1f M ev/s' % (d['value']/1e6, d['ms_per_step'], (d['value_end_to_end'] or 0)/1e6))"
