#!/bin/bash
# Regenerates, on the GPU box, everything kept under profiles/<tag>/: the bench lines of the four
# workloads (one bench run carries cfg2-cfg5), the kernel trace of the default bench, per-workload
# traces and counter passes, and the HBM bytes per path derived from them.
# Results land in gpurun_out/refresh_<tag>/ ; copy what is to be kept into profiles/<tag>/.
set -u
TAG=${1:-r03}
OUT=gpurun_out/refresh_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "[refresh] $*"; }
for wl in cfg2 cfg3 cfg4 cfg5; do
	step "profile $wl"; bash tools/profile.sh ${TAG}x $wl > $OUT/profile_$wl.log 2>&1 || { tail -5 $OUT/profile_$wl.log; exit 1; }
	python3 tools/pmc_summary.py gpurun_out/prof_${TAG}x_$wl > $OUT/${wl}_pmc_summary.json 2>> $OUT/profile_$wl.log
	cp $(find gpurun_out/prof_${TAG}x_$wl/trace -name "*kernel_stats.csv" | head -1) $OUT/${wl}_kernel_stats.csv
done
# the million-object scene once more in the bench's own shape (7680x4320 x 16 spp): L2 hit rates follow the rays' coherence per tile block
step "profile cfg5 at 8K"; bash tools/profile.sh ${TAG}x cfg5 16 7680 4320 _8k > $OUT/profile_cfg5_8k.log 2>&1 || { tail -5 $OUT/profile_cfg5_8k.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/prof_${TAG}x_cfg5_8k > $OUT/cfg5_8k_pmc_summary.json 2>> $OUT/profile_cfg5_8k.log
cp $(find gpurun_out/prof_${TAG}x_cfg5_8k/trace -name "*kernel_stats.csv" | head -1) $OUT/cfg5_8k_kernel_stats.csv
step "L1 counters cfg5"; bash tools/l1_counters.sh ${TAG}x cfg5 > $OUT/l1_cfg5.log 2>&1 || { tail -5 $OUT/l1_cfg5.log; exit 1; }
cp gpurun_out/l1_${TAG}x_cfg5/summary.json $OUT/cfg5_l1_counters.json
python3 tools/traffic_from_pmc.py $OUT $OUT/traffic.json > /dev/null
# the bench line quotes counters and bytes per path from profiles/<tag>/: bring those up to date (in this tree) before it runs
mkdir -p profiles/$TAG
cp $OUT/traffic.json $OUT/*_pmc_summary.json profiles/$TAG/
step "bench cfg2 (default)"; timeout -k 10 500 python3 -u bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -5 $OUT/bench_n1.err; exit 1; }
step "bench cfg2 under rocprofv3 --kernel-trace --stats"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 bench.py --no-cpu --no-extra > $OUT/bench_n1_under_rocprof.json 2> $OUT/bench_trace.err || { tail -5 $OUT/bench_trace.err; exit 1; }
cp $(find $OUT/bench_trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# the raw traces and counter dumps are hundreds of megabytes: only the summaries travel back
rm -rf $OUT/bench_trace gpurun_out/prof_${TAG}x_* gpurun_out/l1_${TAG}x_*
step done; ls $OUT
