export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/jt && mkdir -p /tmp/jt
rocprofv3 --kernel-trace --output-format csv -d /tmp/jt -o jt -- python3 tools/jpeg_timing.py sample-images1 1024 > /tmp/jt/out.txt 2>&1
f=$(find /tmp/jt -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
rows=[r for r in rows if 'jpeg' in r['Kernel_Name'] or 'melf' in r['Kernel_Name']]
t0=min(int(r['Start_Timestamp']) for r in rows)
for r in rows[-40:]:
    print(r.get('Queue_Id'), r.get('Stream_Id'), r['Kernel_Name'][:40], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, r.get('Grid_Size'))
PY
tail -2 /tmp/jt/out.txt
