#!/bin/bash
# Collects the rocprofv3 evidence for one workload into gpurun_out/prof_<tag>/ :
# kernel-trace statistics, then counter passes, each in a run of its own.
set -u
#   bash tools/profile.sh TAG WORKLOAD [SPP WIDTH HEIGHT [SUFFIX]]     (32 spp at 1920x1080 by default)
TAG=${1:-r01}; WL=${2:-cfg2}; SPP=${3:-32}; W=${4:-1920}; H=${5:-1080}; SUFFIX=${6:-}
OUT=gpurun_out/prof_${TAG}_${WL}${SUFFIX}
mkdir -p $OUT
echo "{\"workload\": \"$WL\", \"width\": $W, \"height\": $H, \"spp\": $SPP}" > $OUT/shape.json
export TMPDIR=/tmp
run() { timeout -s KILL 240 rocprofv3 "$@" ; }
run --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/profile_target.py $WL $SPP 3 $W $H > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
run --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -- python3 tools/profile_target.py $WL $SPP 1 $W $H > $OUT/pmc1.log 2>&1 || { echo "pmc1 failed"; tail -5 $OUT/pmc1.log; exit 1; }
run --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc2 -- python3 tools/profile_target.py $WL $SPP 1 $W $H > $OUT/pmc2.log 2>&1 || { echo "pmc2 failed"; tail -5 $OUT/pmc2.log; exit 1; }
run --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 tools/profile_target.py $WL $SPP 1 $W $H > $OUT/pmc3.log 2>&1 || { echo "pmc3 failed"; tail -5 $OUT/pmc3.log; exit 1; }
run --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 tools/profile_target.py $WL $SPP 1 $W $H > $OUT/pmc4.log 2>&1 || { echo "pmc4 failed"; tail -5 $OUT/pmc4.log; exit 1; }
find $OUT -name "*.csv" | head -30
echo done
