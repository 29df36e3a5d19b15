#!/bin/bash
# GPU call r04zd: the driver's round-end sequence in small: smoke(), then the bench with its defaults
O=gpurun_out/r04zd; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 300 python bench.py > $O/bench_defaults.json 2> $O/bench_defaults.err; python -c "
import json; s=open('$O/bench_defaults.json').read(); d=json.loads(s[s.index('{\"metric\"'):]); print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')}); print(d['config']); print(d['roofline']['traffic_stale'], d['roofline']['hbm_bytes_per_step_pmc'], d['cpu_baseline']['value'])"
