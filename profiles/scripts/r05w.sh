#!/bin/bash
# GPU call r05w: the whole GPU suite with poisoned allocations (every fresh device allocation filled with 0xA5: a read of never-written
# memory is deterministic garbage) — the new tables of round 5 (Pc, the voters' records of the elections) included
O=gpurun_out/r05w; mkdir -p $O
(SW_POISON=0xA5 timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_poisoned_allocations.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_poisoned_allocations.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc" $O/pytest_gpu_poisoned_allocations.log | cut -c1-300 | tail -8
