#!/bin/bash
# GPU call r05r: profile recipe at this kernel source (traffic.json), default bench line
O=gpurun_out/r05r; mkdir -p $O; export TMPDIR=/tmp
SW_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) timeout 600 bash profiles/run_profiles.sh r05r > $O/prof.log 2>&1
cp gpurun_out/prof_r05r/traffic.json profiles/traffic.json
head -16 gpurun_out/prof_r05r/kernel_stats.txt
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; s=open('$O/bench_default.json').read(); d=json.loads(s[s.index('{\"metric\"'):]); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['path_frac'], d['find_order_ms'], d['roofline']['traffic_stale'])
for k in d['roofline']['kernels']: print(k['kernel'], k['launches'], k['avg_launch_us'], k['total_ms'], k['frac'], k['hbm_bytes_per_launch_pmc'])"
find gpurun_out/prof_r05r -name '*.db' -size +4M -delete
