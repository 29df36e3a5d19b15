#!/bin/bash
# HEAD verification: the GPU suite with poisoned allocations, the GPU suite as the driver runs it, smoke, bench
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-verify}; mkdir -p $out
SW_POISON=0xA5 timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest_gpu_poison.log 2>&1; echo "rc=$?" >> $out/pytest_gpu_poison.log
grep -E "^FAILED|passed|failed|^rc=|Memory access fault" $out/pytest_gpu_poison.log | cut -c1-300
timeout 500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log
grep -E "^FAILED|passed|failed|^rc=|Memory access fault" $out/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-330 $out/bench.json
