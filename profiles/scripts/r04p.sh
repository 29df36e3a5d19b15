#!/bin/bash
# GPU call r04p: kernel trace of the bench with the early finalize of the last sub-batch (where does it run?)
O=gpurun_out/r04p; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O -o kt -- python bench.py --cpu-sample 0 --e2e-steps 0 --steps 6 --warmup 2 > $O/kt_run.log 2>&1
DB=$(ls $O/*kt_results.db $O/*/*kt_results.db 2>/dev/null | head -1)
python profiles/pass_timeline.py "$DB" 5 > $O/pass_timeline.txt 2>&1
grep -v "k_cansee_fixup\|[345]\.[0-9] us  s.*k_cansee_chunks" $O/pass_timeline.txt | tail -50
tail -2 $O/kt_run.log | cut -c1-400
find $O -name '*.db' -size +8M -delete
