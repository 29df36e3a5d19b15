#!/bin/bash
# BASELINE.json configs[4] shape on ONE GPU: 1024 members, coin-round stress (40 % of the members 50x less
# active), as many events as fit; 256-member / 10 M line for configs[3]
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
C5="--members 1024 --mode 2 --p0 0.40 --p1 0.02 --contexts 1 --cpu-sample 0 --e2e-steps 0 --warmup 0"
for N in 4000000 50000000; do
  timeout 600 python bench.py $C5 --events $N --steps 2 > $O/bench_c5_1024x$N.json 2> $O/c5_$N.err
  python - $O/bench_c5_1024x$N.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(c["members"], c["events"], "%.1f M ev/s %.1f ms | rounds %d coin votes %d (flips %d) | ingest %.2f s" % (d["value"]/1e6, d["ms_per_step"], c["rounds"], c["coin_round_votes"], c["coin_round_votes_from_signature_bit"], c["ingest_s_untimed"]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
  grep -i "Maximum resident\|error\|Error" $O/c5_$N.err | head -3
done
timeout 300 python bench.py --members 1024 --mode 1 --p0 0.002 --contexts 1 --cpu-sample 0 --e2e-steps 0 --warmup 0 --events 20000000 --steps 2 > $O/bench_c5_1024x20M_two_cliques.json 2> $O/c5_cliques.err
python -c "
import json; d=json.load(open('$O/bench_c5_1024x20M_two_cliques.json')); c=d['config']; print('two cliques 1024 x 20M: %.1f M ev/s, rounds %d, coin votes %d (flips %d)' % (d['value']/1e6, c['rounds'], c['coin_round_votes'], c['coin_round_votes_from_signature_bit']))"
