mkdir -p gpurun_out/r6_s3
LF_WATCHDOG=900 timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "border" 2>&1 | tail -5
timeout 1200 python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_s3/c5.json 2> gpurun_out/r6_s3/c5.err
timeout 1200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r6_s3/c2.json 2> gpurun_out/r6_s3/c2.err
python3 - <<'PY'
import json
for f in ('c5','c2'):
    try:
        j=json.loads(open('gpurun_out/r6_s3/%s.json'%f).read().strip().splitlines()[-1])
        bk=j['roofline'].get('by_kernel',{})
        print(f, 'value',round(j['value']), 'ms',round(j['ms_per_step'],1), 'hbm',round(j.get('value_hbm_resident',0)), round(j.get('ms_per_step_hbm_resident',0),1), j['sam_digests']['exclusive_pass']['xxh3_128'][:8], {k.split(' ')[0]:round(v.get('ms_per_step',0),1) for k,v in bk.items()})
    except Exception as e: print(f,'ERR',e)
PY
