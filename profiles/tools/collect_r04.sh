#!/bin/bash
# collect_r04.sh <part> -- round 4's profile sets, run on the GPU box from the repo root; results under gpurun_out/<dir>/ (copied into profiles/<dir>/)
#   part c2      : profiles/tools/collect.sh r04_c2 (bench with the CPU baseline, kernel trace + stats, serialized stats, FETCH / WRITE / SQ counter passes) + the gather microbenchmark
#   part configs : C4 option set on C2's reads, C4 (reads out of segmental duplications, clasp, -n 30, 10 x 100 k), C5, grch38-like repeats
#   part cli     : lordfast --search end to end: plain FASTA / FASTQ, BGZF, one-stream gzip
set -u
PART=${1:-c2}
case $PART in
c2)
  ./profiles/tools/collect.sh r04_c2
  mkdir -p gpurun_out/r04_cu_split
  timeout 300 python3 profiles/tools/gather_tlb_microbench.py > gpurun_out/r04_cu_split/gather_tlb_microbench.txt 2>&1
  tail -12 gpurun_out/r04_cu_split/gather_tlb_microbench.txt
  ;;
configs)
  OUT=$PWD/gpurun_out/r04_configs; mkdir -p $OUT
  timeout 600 python3 bench.py --config c4 --dup-frac 0 --steps 4 --warmup 1 > $OUT/bench_c4_options_on_c2_reads.json 2> $OUT/bench_c4_options.err
  timeout 900 python3 bench.py --config c4 --steps 10 --warmup 1 > $OUT/bench_c4_segdup_clasp_n30.json 2> $OUT/bench_c4.err
  timeout 600 python3 bench.py --config c5 --steps 3 --warmup 1 > $OUT/bench_c5_ont50k_k17c2000.json 2> $OUT/bench_c5.err
  timeout 900 python3 bench.py --repeat-profile grch38like --steps 4 --warmup 1 > $OUT/bench_c2_grch38like.json 2> $OUT/bench_grch38like.err
  for f in $OUT/*.json; do python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('$f'.split('/')[-1], 'value', round(d['value']), round(d['ms_per_step'],1), 'ms; hbm', round(d['value_hbm_resident']), round(d['ms_per_step_hbm_resident'],1), 'ms; match', d.get('primary_record_match_rate'), d.get('all_records_match_rate'), d.get('reads_compared'), d.get('timed_output_equals_exclusive_pass_output'), '; per read', {k: round(v,2) for k,v in d['per_read'].items()})
    print('   exclusive ms', {k.split(' ')[0]: round(v['ms_per_step'],1) for k,v in r['by_kernel'].items()}, 'cpu baseline', round(d.get('cpu_baseline',{}).get('value',0)))
except Exception as e:
    print('$f', 'FAILED', e)
"; done
  ;;
cli)
  OUT=$PWD/gpurun_out/r04_cli; mkdir -p $OUT
  python3 profiles/tools/reader_bench.py 30000 > $OUT/reader_alone_30k_reads.jsonl 2> $OUT/reader_alone.err; cat $OUT/reader_alone_30k_reads.jsonl
  timeout 900 python3 profiles/tools/cli_bench.py --reads 100000 --copies 3 --devnull-only > $OUT/cli_fasta_300k_reads.jsonl 2> $OUT/cli_fasta.err; cat $OUT/cli_fasta_300k_reads.jsonl
  timeout 900 python3 profiles/tools/cli_bench.py --reads 100000 --copies 2 --fastq --devnull-only > $OUT/cli_fastq_200k_reads.jsonl 2> $OUT/cli_fastq.err; cat $OUT/cli_fastq_200k_reads.jsonl
  timeout 900 python3 profiles/tools/cli_bench.py --reads 100000 --copies 2 --fastq --bgzf --devnull-only > $OUT/cli_fastq_bgzf_200k_reads.jsonl 2> $OUT/cli_fastq_bgzf.err; cat $OUT/cli_fastq_bgzf_200k_reads.jsonl
  LF_TIMING=1 timeout 900 python3 profiles/tools/cli_bench.py --reads 100000 --fastq --gz --devnull-only > $OUT/cli_fastq_gz_100k_reads.jsonl 2> $OUT/cli_fastq_gz.err; cat $OUT/cli_fastq_gz_100k_reads.jsonl; grep "CPU-s in inflate" $OUT/cli_fastq_gz.err | tail -2
  ;;
esac
