#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-forks2}; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_forks.py tests/test_gpu_parity.py tests/test_gpu_node.py -x -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log | cut -c1-300
timeout 100 python profiles/exact_bench.py > $out/exact_bench.log 2>&1; tail -3 $out/exact_bench.log
