# per-kernel durations of one frame rendered alone (rocprofv3 kernel trace of bench.py --inflight 1)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
rm -rf $R/gpurun_out/lv
RT_TRACE_LEVELS=1 timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lv -- python3 $R/bench.py --steps 3 --warmup 1 --inflight 1 --sustain-s 0 --hw-queues 0 --no-cpu-baseline "$@" > $R/gpurun_out/lv.json 2> $R/gpurun_out/lv.err || { tail -5 $R/gpurun_out/lv.err; exit 1; }
grep "rt: level" $R/gpurun_out/lv.err
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/lv/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_wf' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last frame: from the last primary-level launch on
last=max(i for i,r in enumerate(rows) if 'k_wf_level<false, true' in r['Kernel_Name'])
t0=int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    print('%-44s start %8.1f us  dur %8.1f us' % (r['Kernel_Name'][:44], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
