#!/bin/bash
# round 3: where the hot-member hashgraph goes (13 of 256 members create 96 % of the events)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
timeout 300 python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 --mode 2 --p0 0.95 --p1 0.002 > $O/bench_hot.json 2> $O/bench_hot.err
python -c "
import json; d=json.load(open('$O/bench_hot.json')); print('hot', d['value'], d['ms_per_step'], d['roofline']['counters'], d['roofline']['phase_ms'])
for k in d['roofline']['kernels']: print(k['kernel'], k['launches'], k['avg_launch_us'], k['total_ms'])"
rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --contexts 1 --mode 2 --p0 0.95 --p1 0.002 > $O/kt.log 2>&1
DB=$(ls $O/kt/*kt_results.db $O/kt/*/*kt_results.db 2>/dev/null | head -1)
python profiles/summarize_rocpd.py "$DB" > $O/kernel_stats_hot.txt 2>> $O/kt.log
python profiles/loop_timeline.py "$DB" > $O/loop_timeline_hot.txt 2>> $O/kt.log
head -12 $O/kernel_stats_hot.txt; cat $O/loop_timeline_hot.txt
find $O -name '*.db' -size +8M -delete
