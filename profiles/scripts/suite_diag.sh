#!/bin/bash
# full GPU suite (no -x), then the windowed-table tests alone under host CPU contention
cd "$GRAFT_REPO_ROOT" || exit 1
python py-swirld_amd/build.py --force > /dev/null 2>&1
out=gpurun_out/${1:-diag}; mkdir -p $out
nproc > $out/nproc.txt
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/suite.log 2>&1; echo "rc=$?" >> $out/suite.log
tail -5 $out/suite.log | cut -c1-600
# CPU burners: one busy loop per core
python - <<'PY' &
import multiprocessing as mp, time, os
def burn(t):
    e = time.time() + t
    x = 0
    while time.time() < e:
        x += 1
if __name__ == "__main__":
    n = os.cpu_count()
    ps = [mp.Process(target=burn, args=(150,)) for _ in range(n)]
    [p.start() for p in ps]
    [p.join() for p in ps]
PY
BURN=$!
for i in 1 2 3; do
  timeout 120 python -m pytest tests/test_gpu_window.py -m gpu -q -x -p no:cacheprovider -k "24-150000" > $out/burn_$i.log 2>&1; echo "rc=$?" >> $out/burn_$i.log
  tail -3 $out/burn_$i.log | cut -c1-600
done
kill $BURN 2>/dev/null; wait $BURN 2>/dev/null
