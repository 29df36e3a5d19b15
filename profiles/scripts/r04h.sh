#!/bin/bash
# GPU call r04h: the resolve step timed without the per-phase stamps; bench line with the two-contexts-at-once figure;
# the profile recipe (kernel trace, PMC passes, traffic.json) on this tree's kernels
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
SW_DEBUG_CLOCKS=2 timeout 100 python profiles/resolve_time.py 256 1000000 > $O/resolve_time.txt 2>&1
SW_DEBUG_CLOCKS=1 timeout 100 python profiles/resolve_time.py 256 1000000 >> $O/resolve_time.txt 2>&1
cat $O/resolve_time.txt
timeout 300 python bench.py --steps 12 --warmup 3 --cpu-sample 0 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['value_concurrent_contexts'], d['find_order_ms'], d['find_order_first_call_ms'])"
SW_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) timeout 600 bash profiles/run_profiles.sh r04h > $O/prof.log 2>&1
head -14 gpurun_out/prof_r04h/kernel_stats.txt; cat gpurun_out/prof_r04h/loop_timeline.txt | head -6; cat gpurun_out/prof_r04h/traffic.json | head -c 600
