#!/bin/bash
# round 3: where 1024 members go (phase stamps + kernel trace), small calls after the loop changes
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; tail -5 $O/build.log; exit 1; }
SW_DEBUG_CLOCKS=1 timeout 200 python profiles/loop_phases.py 1024 2000000 > $O/loop_phases_1024.txt 2>&1
head -45 $O/loop_phases_1024.txt
rocprofv3 --kernel-trace -d $O/kt1024 -o kt -- python bench.py --cpu-sample 0 --e2e-steps 0 --steps 3 --warmup 1 --members 1024 --events 2000000 > $O/kt1024.log 2>&1
DB=$(ls $O/kt1024/*kt_results.db $O/kt1024/*/*kt_results.db 2>/dev/null | head -1)
python profiles/summarize_rocpd.py "$DB" > $O/kernel_stats_1024x2M.txt 2>> $O/kt1024.log
python profiles/loop_timeline.py "$DB" > $O/loop_timeline_1024x2M.txt 2>> $O/kt1024.log
head -14 $O/kernel_stats_1024x2M.txt; cat $O/loop_timeline_1024x2M.txt
find $O -name '*.db' -size +8M -delete
timeout 300 python profiles/incremental_bench.py > $O/incremental_small_batches.log 2>&1; tail -8 $O/incremental_small_batches.log | cut -c1-400
