mkdir -p gpurun_out
( timeout -k 10 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail=8 ) > gpurun_out/pytest2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest2.log
tail -25 gpurun_out/pytest2.log
for cfg in "1 1 16" "1 1 32" "1 0 16" "0 1 16"; do set -- $cfg
  echo "== cansee=$1 tally=$2 K=$3" >> gpurun_out/bench2.log
  ( SW_CANSEE_IMPL=$1 SW_TALLY_IMPL=$2 SW_TALLY_K=$3 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 ) >> gpurun_out/bench2.log 2>&1
done
( timeout -k 10 300 python bench.py --steps 3 --warmup 1 --members 64 --events 100000 --cpu-sample 0 ) >> gpurun_out/bench2.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench2.log'):
    if l.startswith('=='): print(l.strip())
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['config']['members'], d['value'], d['ms_per_step'], r['phase_ms'], r['avg_launch_us'], r['launches'], r['evals_per_launch'])
PY
