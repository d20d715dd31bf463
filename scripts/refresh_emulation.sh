#!/bin/bash
# profiles/r03/sharded_emulation_*.log: scripts/emulate_sharded.py at 1M and 10M, both datasets (one gpurun call)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/emu
mkdir -p $OUT
HOST_US_PER_COLLECTIVE=60 python3 $ROOT/scripts/emulate_sharded.py 1000000 768 survey > $OUT/sharded_emulation_1m_survey.log 2>&1
echo "1m survey done"; tail -1 $OUT/sharded_emulation_1m_survey.log | cut -c1-200
HOST_US_PER_COLLECTIVE=60 python3 $ROOT/scripts/emulate_sharded.py 1000000 768 tight > $OUT/sharded_emulation_1m_tight.log 2>&1
echo "1m tight done"
EMU_WORLDS=1,8 HOST_US_PER_COLLECTIVE=60 python3 $ROOT/scripts/emulate_sharded.py 10000000 768 survey > $OUT/sharded_emulation_10m_survey.log 2>&1
echo "10m survey done"; tail -1 $OUT/sharded_emulation_10m_survey.log | cut -c1-200
EMU_WORLDS=1,8 HOST_US_PER_COLLECTIVE=60 python3 $ROOT/scripts/emulate_sharded.py 10000000 768 tight > $OUT/sharded_emulation_10m_tight.log 2>&1
echo "10m tight done"; tail -1 $OUT/sharded_emulation_10m_tight.log | cut -c1-200
