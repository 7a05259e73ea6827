# the whole GPU suite, then C2 / C5 / C4 quick lines
mkdir -p gpurun_out/r6_full
timeout 3000 python3 -m pytest tests/ -x -q -m gpu 2>&1 | tail -5
for cfg in c2 c5 c4; do
if [ $cfg = c2 ]; then A="--steps 6 --warmup 2"; else A="--config $cfg --steps 2 --warmup 1"; fi
for i in 1 2; do
f=gpurun_out/r6_full/${cfg}_$i.json
timeout 900 python3 bench.py $A --no-cpu-baseline --no-host-region > $f 2> gpurun_out/r6_full/err.txt
python3 - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
bk=d["roofline"]["by_kernel"]
print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],1), {k.split(' ')[0]:round(v["ms_per_step"],1) for k,v in bk.items() if "hirsch" in k or "rsweep" in k or "tb_k" in k}, d.get("timed_output_equals_exclusive_pass_output"))
PY
done
done
