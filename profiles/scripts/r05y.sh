#!/bin/bash
# GPU call r05y: HEAD — whole GPU suite, smoke, default bench line
O=gpurun_out/r05y; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu.log | cut -c1-300 | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; s=open('$O/bench_default.json').read(); d=json.loads(s[s.index('{\"metric\"'):]); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['path_frac'], d['roofline']['traffic_stale'], d['cpu_baseline']['value'])"
