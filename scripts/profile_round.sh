#!/bin/bash
# rocprofv3 passes behind profiles/<round>/ (run on the GPU box through gpurun):
#   scripts/profile_round.sh r01
# kernel trace + stats in one pass, the two PMC counters in passes of their own.
set -e
ROUND=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$ROUND
FLAGS="--skip-iid --skip-pq --cpu-seconds 0 --ef 104 --probe-depth 8"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --steps 20 $FLAGS > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace pass done" && tail -c 300 $OUT/bench_trace.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 5 --warmup 1 $FLAGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --steps 5 --warmup 1 $FLAGS > $OUT/bench_write.json 2> $OUT/bench_write.err
echo "write pass done"
# the raw per-dispatch CSVs are large; keep the summaries
python3 $ROOT/scripts/make_profile_summary.py $OUT $OUT/summary
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
# index construction alone (1M x 768 clustered, single GPU): per-kernel totals
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/build -o b -- python3 $ROOT/scripts/probe_plain_build.py > $OUT/build.log 2>&1
cp $OUT/build/*kernel_stats.csv $OUT/summary/kernel_stats_build.csv 2>/dev/null || find $OUT/build -name "*kernel_stats.csv" -exec cp {} $OUT/summary/kernel_stats_build.csv \;
find $OUT -name "*kernel_trace.csv" -size +8M -delete
tail -1 $OUT/build.log
