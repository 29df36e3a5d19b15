#!/bin/bash
# SW_POISON runs: fresh device memory filled with a byte pattern, so reads of never-written memory show up
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-poison}; mkdir -p $out
for b in 0xA5 0x7F 0xFF; do
  SW_POISON=$b timeout 400 python -m pytest tests/test_gpu_window.py tests/test_gpu_parity.py tests/test_gpu_ingest.py -m gpu -x -q -s -p no:cacheprovider > $out/poison_$b.log 2>&1
  echo "rc=$?" >> $out/poison_$b.log
  echo "== $b"; tail -4 $out/poison_$b.log | cut -c1-400
done
