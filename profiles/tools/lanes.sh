#!/bin/bash
# lanes.sh -- bench at several chunk-in-flight settings (same box, back to back) + host phase accounting
mkdir -p gpurun_out/lanes
for L in 8 12 16; do
  LF_LANES=$L python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/lanes/l$L.json 2> gpurun_out/lanes/l$L.err
  python3 -c "import json;d=json.load(open('gpurun_out/lanes/l$L.json'));print('lanes',$L,round(d['value']),round(d['ms_per_step'],1),d['host_cpu_seconds_per_step'])"
done
LF_PHASES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/lanes/ph.json 2> gpurun_out/lanes/ph.err
grep "phase\|batch of" gpurun_out/lanes/ph.err | tail -32
LF_LANES=1 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/lanes/one.json 2> gpurun_out/lanes/one.err
python3 -c "
import json;d=json.load(open('gpurun_out/lanes/one.json'));print('one lane',round(d['value']),round(d['ms_per_step'],1));print({k:round(v['ms_per_step'],1) for k,v in d['roofline']['by_kernel'].items()})"
