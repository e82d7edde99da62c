export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
RT_TRACE_LEVELS=1 timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lv -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > $R/gpurun_out/lv.json 2> $R/gpurun_out/lv.err
grep "rt: level" $R/gpurun_out/lv.err
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/lv/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_wf' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-9:]:
    print(r['Kernel_Name'][:40], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us')
PY
