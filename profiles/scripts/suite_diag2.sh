#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-diag}; mkdir -p $out
timeout 400 python -m pytest tests -m "gpu and not heavy" -q -p no:cacheprovider > $out/suite_a.log 2>&1; echo "rc=$?" >> $out/suite_a.log
grep -E "Failed:|passed|failed" $out/suite_a.log | cut -c1-1200
SW_POISON=0xA5 timeout 400 python -m pytest tests -m "gpu and not heavy" -q -p no:cacheprovider > $out/suite_b.log 2>&1; echo "rc=$?" >> $out/suite_b.log
grep -E "Failed:|passed|failed" $out/suite_b.log | cut -c1-1200
