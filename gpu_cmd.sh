mkdir -p gpurun_out
for cfg in "1 4" "0 4" "1 8" "0 8" "1 2"; do set -- $cfg
  echo "== prio=$1 pipe=$2" >> gpurun_out/bench14.log
  ( SW_PRIO=$1 SW_PIPE=$2 timeout -k 10 300 python bench.py --steps 6 --warmup 2 --cpu-sample 0 ) >> gpurun_out/bench14.log 2>&1
done
python - <<'PY'
import json
for l in open('gpurun_out/bench14.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['phase_ms'], r['avg_launch_us'], r['launches'])
    elif 'amdgpu.ids' not in l: print(l.strip()[:300])
PY
