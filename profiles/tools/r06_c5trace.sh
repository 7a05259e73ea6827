# r06_c5trace.sh [c5|c4]: the Hirschberg launches of the last step in the exclusive pass's mode (one lane, one chunk, the queues one after the other), in time order
CFG=${1:-c5}
mkdir -p gpurun_out/r6_c5trace gpurun_out/r06_hirsch
cd /tmp && export TMPDIR=/tmp
export LF_LANES=1 LF_SERIAL_CLASSES=1 LF_CHUNK_READS=1073741824 LF_CHUNK_BASES=1099511627776
if [ $CFG = c4 ]; then export LF_CHUNK_READS=25000; fi      # (-n 30 on duplicated reads: bench.py's exclusive pass keeps 25 k-read chunks too)
for band in 1; do
LF_HIRSCH_BAND=$band rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6_c5trace/ser_band$band -o c5 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > $GRAFT_REPO_ROOT/gpurun_out/r6_c5trace/c5_ser_band$band.json 2> $GRAFT_REPO_ROOT/gpurun_out/r6_c5trace/err_band$band.txt
python3 - $(find $GRAFT_REPO_ROOT/gpurun_out/r6_c5trace/ser_band$band -name "*kernel_trace.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/r06_hirsch/${CFG}_last_step_launches_alone_on_the_gpu.txt <<'PY'
import csv,sys,re
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'hband' in r['Kernel_Name'] or 'hirsch' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
roots=[i for i,r in enumerate(rows) if 'roots' in r['Kernel_Name']]
half=roots[len(roots)//2]
tot={}
prev_end=None
for r in rows[half:]:
    nm=re.sub(r'void |\(lf_hargs.*|lf_', '', r['Kernel_Name'])
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    wg=int(r['Workgroup_Size_X']); gr=int(r['Grid_Size_X'])//wg
    gap=(st-prev_end)/1e6 if prev_end else 0
    print(f"{nm:40s} workgroups {gr:7d} x {wg:4d}  {(en-st)/1e6:8.2f} ms   gap before {gap:7.2f} ms")
    tot[nm]=tot.get(nm,0)+(en-st)/1e6
    prev_end=en
print({k:round(v,1) for k,v in sorted(tot.items(), key=lambda x:-x[1])}, round(sum(tot.values()),1))
PY
find $GRAFT_REPO_ROOT/gpurun_out/r6_c5trace/ser_band$band -name "*kernel_trace.csv" -delete
done
