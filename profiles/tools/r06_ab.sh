# A / B of the Hirschberg levels on C4 and C5 (T2T-like): banded (default) against LF_HIRSCH_BAND=0, alternating on one box
mkdir -p gpurun_out/r6_ab
for rep in 1 2; do for band in 1 0; do
  LF_HIRSCH_BAND=$band timeout 900 python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive > gpurun_out/r6_ab/c4_band${band}_$rep.json 2> gpurun_out/r6_ab/c4_band${band}_$rep.err
done; done
for band in 1 0; do
  LF_HIRSCH_BAND=$band timeout 1200 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ab/c5_band${band}.json 2> gpurun_out/r6_ab/c5_band${band}.err
done
LF_HIRSCH_BAND=1 timeout 900 python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_ab/c4_excl_band1.json 2> gpurun_out/r6_ab/c4_excl_band1.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_ab/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        bk=j['roofline'].get('by_kernel',{})
        print(f.split('/')[-1], 'value',round(j['value']), 'ms',round(j['ms_per_step'],1), 'hbm',round(j.get('value_hbm_resident',0)), round(j.get('ms_per_step_hbm_resident',0),1), j['sam_digests']['hbm_resident_timed_steps']['xxh3_128'][:8], {k.split(' ')[0]:round(v.get('ms_per_step',0),1) for k,v in bk.items() if v.get('ms_per_step',0)>20})
    except Exception as e: print(f,'ERR',e)
PY
