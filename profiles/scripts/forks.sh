#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-forks}; mkdir -p $out
timeout 420 python -m pytest tests/test_gpu_forks.py tests/test_gpu_parity.py::test_fork_is_refused tests/test_gpu_ingest.py::test_bulk_append_is_atomic_on_rejection tests/test_gpu_node.py -x -q -p no:cacheprovider --durations=8 > $out/pytest_forks.log 2>&1; echo "rc=$?" >> $out/pytest_forks.log
tail -40 $out/pytest_forks.log | cut -c1-300
