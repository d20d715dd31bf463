#!/bin/bash
# rocprofv3 passes behind profiles/r03/ (run on the GPU box through gpurun): scripts/profile_round3.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (a) kernel trace + stats of the measuring worker itself (headline only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --role worker --steps 20 --skip-iid --skip-tight --skip-pq --skip-sharded-build --cpu-seconds 0 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
echo "trace pass done"
# (b) the plain run: driver + worker + its own --pmc passes (search: read / write; build: read + kernel trace); counter CSVs kept
python3 $ROOT/bench.py --keep-pmc $OUT/pmc > $OUT/bench_full.json 2> $OUT/bench_full.err
echo "plain run done"
python3 $ROOT/scripts/make_profile_summary_r3.py $OUT $OUT/summary
# (c) index construction alone (1M x 768, SURVEY clustered variant, single GPU): per-kernel totals
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/build -o b -- python3 $ROOT/scripts/probe_plain_build.py > $OUT/build.log 2>&1
find $OUT/build -name "*kernel_stats.csv" -exec cp {} $OUT/summary/kernel_stats_build.csv \;
# (d) the PQ search kernel (config 5): SQ, LDS and L1->L2 counters
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pq_a -o a -- python3 $ROOT/scripts/probe_one.py pq 448 8 > $OUT/pq_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pq_b -o b -- python3 $ROOT/scripts/probe_one.py pq 448 8 > $OUT/pq_b.log 2>&1
python3 $ROOT/scripts/pmc_kernel.py $OUT/pq_a ph_search_kernel 2 > $OUT/summary/pq_counters.txt
python3 $ROOT/scripts/pmc_kernel.py $OUT/pq_b ph_search_kernel 2 >> $OUT/summary/pq_counters.txt
grep "^pq " $OUT/pq_a.log >> $OUT/summary/pq_counters.txt
# (e) the shape of a launch: fixed cost + marginal rate (isolated launches of 3 072 ... 32 000 queries), and the host path
python3 $ROOT/scripts/probe_tail.py 2>&1 | grep "^nq" > $OUT/summary/launch_size_sweep.txt
python3 $ROOT/scripts/probe_host_path.py 2>&1 | grep "^nq" > $OUT/summary/host_path.txt
find $OUT -name "*kernel_trace.csv" -size +8M -delete
cat $OUT/summary/pq_counters.txt $OUT/summary/launch_size_sweep.txt $OUT/summary/host_path.txt
