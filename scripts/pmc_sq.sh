#!/bin/bash
# SQ counters of the search kernel for one configuration: scripts/pmc_sq.sh <tag> <probe_one args...>
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -o a -- python3 $ROOT/scripts/probe_one.py "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/b -o b -- python3 $ROOT/scripts/probe_one.py "$@" > $OUT/b.log 2>&1
tail -1 $OUT/a.log
python3 $ROOT/scripts/pmc_kernel.py $OUT/a ph_search_kernel 2
python3 $ROOT/scripts/pmc_kernel.py $OUT/b ph_search_kernel 2
