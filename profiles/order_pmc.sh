#!/bin/bash
# PMC passes over find_order (profiles/order_laps.py): HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), L2 hits / misses, wave waits.
#   profiles/order_pmc.sh <tag> [members events]   -> gpurun_out/<tag>/order_pmc.txt
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for pass in "f FETCH_SIZE" "w WRITE_SIZE" "t TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "s SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
    set -- $pass
    name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- python profiles/order_laps.py ${ORDER_ARGS:-} > $OUT/pmc_$name.log 2>&1
done
python profiles/summarize_pmc_csv.py $OUT/pmc_f $OUT/pmc_w $OUT/pmc_t $OUT/pmc_s 2>&1 | grep "^#\|k_order\|dispatches" > $OUT/order_pmc.txt
find $OUT -name '*.csv' -size +2M -delete
find $OUT -name '*.db' -delete
