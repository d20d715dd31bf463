set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/table_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/scripts/probe_table.py > $OUT/trace.log 2>&1
grep "^table" $OUT/trace.log
find $OUT/trace -name "*kernel_stats.csv" -exec grep -h "tiny" {} \; | cut -c1-200
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o a -- python3 $GRAFT_REPO_ROOT/scripts/probe_table.py > $OUT/a.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_kernel.py $OUT/a ph_tiny_table_mfma 2
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -o b -- python3 $GRAFT_REPO_ROOT/scripts/probe_table.py > $OUT/b.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_kernel.py $OUT/b ph_tiny_table_mfma 2
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -o c -- python3 $GRAFT_REPO_ROOT/scripts/probe_table.py > $OUT/c.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_kernel.py $OUT/c ph_tiny_table_mfma 2
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +4M -delete
