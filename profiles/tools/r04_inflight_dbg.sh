#!/bin/bash
OUT=gpurun_out/${1:-r04p}; mkdir -p $OUT
B="python3 -X faulthandler bench.py --no-cpu-baseline --no-exclusive --no-host-region --reads 12500 --steps 8 --warmup 1"
timeout 200 $B --inflight 2 > $OUT/d2.json 2> $OUT/d2.err; echo "rc $?"; grep -v "^\[bench\]\|amdgpu" $OUT/d2.err | head -40
if [ ! -s $OUT/d2.json ]; then
  timeout 300 /opt/rocm/bin/rocgdb -batch -ex run -ex bt -ex "thread apply all bt 12" --args $B --inflight 2 > $OUT/gdb.txt 2>&1
  grep -n "SIGSEGV\|SIGABRT\|signal" -A 30 $OUT/gdb.txt | head -80
fi
