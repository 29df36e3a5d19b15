#!/bin/bash
# GPU call r04v: the default bench line with the traffic.json of this code in the tree; the whole GPU suite with poisoned allocations
O=gpurun_out/r04v; mkdir -p $O
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; s=open('$O/bench_default.json').read(); d=json.loads(s[s.index('{\"metric\"'):]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['traffic_stale'], r['hbm_bytes_per_step_pmc'], r['frac_hbm_measured'], r['traffic'])"
(SW_POISON=165 timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_poisoned_allocations.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_poisoned_allocations.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu_poisoned_allocations.log | cut -c1-300 | tail -8
