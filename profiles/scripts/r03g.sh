#!/bin/bash
# round 3: the whole GPU suite at HEAD, the default bench line, 1024 x 2 M and 64 x 100 k lines
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu.log | cut -c1-300 | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['path_frac'])
for k in d['roofline']['kernels']: print(k['kernel'], k['launches'], k['avg_launch_us'], k['total_ms'], k['frac'])"
timeout 200 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 5 --warmup 2 --members 1024 --events 2000000 > $O/bench_1024x2M.json 2>> $O/err.log
timeout 200 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 10 --warmup 2 --members 64 --events 100000 > $O/bench_64x100k.json 2>> $O/err.log
python -c "
import json
for f in ('bench_1024x2M','bench_64x100k'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'])"
