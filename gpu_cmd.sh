mkdir -p gpurun_out/prof_r01b
R=$GRAFT_REPO_ROOT
( timeout -k 10 1500 python -m pytest tests -m gpu -q --timeout 900 --maxfail=8 ) > gpurun_out/pytest11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest11.log
tail -6 gpurun_out/pytest11.log
for cfg in "8" "4"; do
  echo "== pipe=$cfg" >> gpurun_out/bench11.log
  ( SW_PIPE=$cfg timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-sample 0 ) >> gpurun_out/bench11.log 2>&1
done
echo "== 64/100k" >> gpurun_out/bench11.log
( timeout -k 10 300 python bench.py --steps 5 --warmup 1 --members 64 --events 100000 --cpu-sample 0 ) >> gpurun_out/bench11.log 2>&1
echo "== 256/10M" >> gpurun_out/bench11.log
( timeout -k 10 600 python bench.py --steps 2 --warmup 1 --events 10000000 --contexts 1 --cpu-sample 0 ) >> gpurun_out/bench11.log 2>&1
echo "== 1024/2M" >> gpurun_out/bench11.log
( timeout -k 10 600 python bench.py --steps 2 --warmup 1 --members 1024 --events 2000000 --contexts 1 --cpu-sample 0 ) >> gpurun_out/bench11.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/bench11.log'):
    if l.startswith('=='): print(l.strip())
    elif l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['config']['members'], d['config']['events'], d['value'], d['ms_per_step'], r['phase_ms'], r['avg_launch_us'], r['launches'], r['evals_per_launch'], r.get('far_hops'), d['config']['rounds'])
    elif 'amdgpu.ids' not in l: print(l.strip()[:300])
PY
cd /tmp && export TMPDIR=/tmp
( timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01b -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 ) > $R/gpurun_out/prof_r01b/kt_run.log 2>&1
cd $R
python profiles/summarize_rocpd.py gpurun_out/prof_r01b/kt_results.db > gpurun_out/prof_r01b/kernel_stats.txt 2>&1
find gpurun_out/prof_r01b -name "*.db" -delete
head -8 gpurun_out/prof_r01b/kernel_stats.txt | cut -c1-60,73-200
grep "^{" gpurun_out/prof_r01b/kt_run.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['launches'])"
