# the matrix-core table kernel alone: durations of its last three launches from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_t -o t -- python3 $GRAFT_REPO_ROOT/scripts/probe_table.py > /tmp/tr_t.log 2>&1
grep '^table' /tmp/tr_t.log
python3 - <<PY
import csv,glob
f=glob.glob('/tmp/tr_t/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'ph_tiny_table_mfma' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in rows[-3:]]
print('last 3 table kernels ms', d, 'grid', rows[-1].get('Grid_Size_X') or rows[-1].get('Grid_Size'))
PY
