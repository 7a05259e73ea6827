#!/bin/bash
# configs.sh -- bench lines of the other BASELINE configs' option sets and of the repeat-rich genome profile (builder-run
# evidence; the driver's headline stays config C2 on the default genome)
OUT=$PWD/gpurun_out/${1:-r03_configs}
mkdir -p $OUT
python3 bench.py --config c4 --steps 2 --warmup 1 > $OUT/bench_c4_clasp_n30.json 2> $OUT/bench_c4.err
python3 bench.py --config c5 --steps 2 --warmup 1 > $OUT/bench_c5_ont50k_k17c2000.json 2> $OUT/bench_c5.err
python3 bench.py --repeat-profile grch38like --steps 2 --warmup 1 > $OUT/bench_c2_grch38like.json 2> $OUT/bench_grch38like.err
for f in $OUT/*.json; do python3 -c "
import json,sys
d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], round(d['value']), 'reads/s', round(d['ms_per_step'],1), 'ms; match', d.get('primary_record_match_rate'), d.get('all_records_match_rate'), d.get('reads_compared'), '; per read', {k: round(v,1) for k,v in d['per_read'].items()})
print('   exclusive ms', {k: round(v['ms_per_step'],1) for k,v in r['by_kernel'].items()})"; done
