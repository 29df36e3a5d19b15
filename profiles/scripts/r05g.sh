#!/bin/bash
# GPU call r05g: anatomy of a pass at the current code (untraced per-sub-batch times; kernel trace -> pass timeline, chained and not)
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python profiles/subbatch_times.py > $O/subbatch_times_256x1M.txt 2>&1; tail -22 $O/subbatch_times_256x1M.txt
for ch in 1 0; do
SW_CHAIN=$ch timeout 300 rocprofv3 --kernel-trace -d $O/kt$ch -o kt -- python bench.py --steps 8 --warmup 3 --cpu-sample 0 --e2e-steps 0 --contexts 1 > $O/kt${ch}_run.log 2>&1
DB=$(ls $O/kt$ch/*kt_results.db $O/kt$ch/*/*kt_results.db 2>/dev/null | head -1)
python profiles/pass_timeline.py "$DB" 6 > $O/pass_timeline_chain$ch.txt 2>&1
python profiles/summarize_rocpd.py "$DB" > $O/kernel_stats_chain$ch.txt 2>&1
done
find $O -name '*.db' -size +8M -delete
cat $O/pass_timeline_chain1.txt | head -80
