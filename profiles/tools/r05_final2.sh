OUT=$PWD/gpurun_out/r05_final; mkdir -p $OUT
for k in 1 2; do timeout 900 python3 bench.py > $OUT/bench_default_run$k.json 2> $OUT/bench_default_run$k.err; python3 - $OUT/bench_default_run$k.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('default bench: value',round(d['value']),round(d['ms_per_step'],1),'ms; hbm',round(d['value_hbm_resident']),round(d['ms_per_step_hbm_resident'],1),'; cpu',d['host_cpu_seconds_per_step'],d['host_cpu_seconds_per_step_hbm_resident'],'; match',d.get('all_records_match_rate'),d.get('reads_compared'),'; roofline',r['kernel'],round(r['frac'],4),'traffic',r['traffic'],'alu',(r.get('alu') or {}).get('frac'),'cpu baseline',d['cpu_baseline']['value'], 'sha', d.get('bench_py_sha16'), d.get('source_tree'))
PY
done
python3 bench.py --tree-hash > $OUT/SOURCE_TREE.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
