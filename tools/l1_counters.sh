#!/bin/bash
# L1 (TCP/TA) counters of the render kernel on one workload, two passes:
#   bash tools/l1_counters.sh <tag> <workload>     -> gpurun_out/l1_<tag>_<workload>/
set -u
TAG=${1:-r01}; WL=${2:-cfg5}
OUT=gpurun_out/l1_${TAG}_${WL}
mkdir -p $OUT
export TMPDIR=/tmp
run() { timeout -s KILL 120 rocprofv3 "$@" ; }
run --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1 -- python3 tools/profile_target.py $WL 32 1 > $OUT/pmc1.log 2>&1 || { echo "pass 1 failed"; tail -5 $OUT/pmc1.log; exit 1; }
run --pmc TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/pmc2 -- python3 tools/profile_target.py $WL 32 1 > $OUT/pmc2.log 2>&1 || { echo "pass 2 failed"; tail -5 $OUT/pmc2.log; exit 1; }
python3 tools/pmc_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
