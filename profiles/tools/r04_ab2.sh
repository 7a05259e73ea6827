#!/bin/bash
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_map.py -x -q -m gpu -k "pinned or caller_buffer or fastq" > $OUT/pytest_pinned.log 2>&1; tail -3 $OUT/pytest_pinned.log
for i in 1 2; do for C in default 8334 12500; do
  if [ $C = default ]; then unset LF_CHUNK_READS; else export LF_CHUNK_READS=$C; fi
  timeout 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-exclusive 2>$OUT/ab_$C_$i.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $C', 'value', round(d['value']), round(d['ms_per_step'],2), 'hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],2), 'cpu', round(d['host_cpu_seconds_per_step'],3))"
done; done
