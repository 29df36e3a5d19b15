#!/bin/bash
# last evidence of round 3 at HEAD: whole GPU suite, default bench line, profile recipe
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
(timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu.log | cut -c1-300 | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['value_end_to_end'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['path_frac'])
for k in d['roofline']['kernels']: print(k['kernel'], k['launches'], k['avg_launch_us'], k['total_ms'], k['frac'], k['hbm_bytes_per_launch_pmc'])"
SW_COMMIT=$(cat .commit_id 2>/dev/null || echo unknown) timeout 600 bash profiles/run_profiles.sh $1 > $O/prof.log 2>&1
head -8 gpurun_out/prof_$1/kernel_stats.txt; cat gpurun_out/prof_$1/loop_timeline.txt | head -6
SW_DEBUG_CLOCKS=1 SW_PIPE=1 timeout 120 python profiles/loop_phases.py 256 1000000 > $O/loop_phases_256x1M.txt 2>&1
grep -E "iteration period|end - entry|hop masks" $O/loop_phases_256x1M.txt
