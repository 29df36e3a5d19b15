#!/bin/bash
# the whole GPU suite with poisoned allocations (fresh device memory filled with 0xA5): reads of never-written memory become deterministic
O=gpurun_out/$1; mkdir -p $O
python py-swirld_amd/build.py --force > $O/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
(SW_POISON=0xA5 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_poisoned.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_poisoned.log)
grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|pytest rc|Error" $O/pytest_gpu_poisoned.log | cut -c1-300 | tail -8
