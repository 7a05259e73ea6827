#!/bin/bash
# r05_hist.sh <tag> -- where the class-1 alignment problems of a C2 step are: LF_HIST_STATS=1 prints the counting sort's bins (mode, blocks,
# target-length bucket) per alignment round; summed here over one 20 k-read step (problems, block steps, tiles by blocks per problem)
OUT=gpurun_out/${1:-r05hist}; mkdir -p $OUT
LF_HIST_STATS=1 timeout 900 python3 bench.py --reads 20000 --steps 1 --warmup 0 --no-cpu-baseline --no-host-region --no-exclusive > $OUT/b.json 2> $OUT/b.err
python3 - $OUT/b.err > $OUT/hist.txt <<'PY'
import re,sys,collections
cnt=collections.Counter()
for l in open(sys.argv[1]):
    m=re.match(r"\[lf\] bin mode (\d+) nb (\d+) mb (\d+): (\d+)", l)
    if m: cnt[(int(m.group(1)),int(m.group(2)),int(m.group(3)))]+=int(m.group(4))
def mid(mb): return mb*32+16 if mb<16 else 512+(mb-16)*256+128
tot=sum(cnt.values()); steps=sum(c*nb*mid(mb) for (mo,nb,mb),c in cnt.items()); tiles=sum(c*(mid(mb)/16+nb) for (mo,nb,mb),c in cnt.items())
print("problems",tot,"block steps %.3g"%steps,"tb tiles %.3g"%tiles)
bynb=collections.defaultdict(lambda:[0,0,0])
for (mo,nb,mb),c in cnt.items():
    k=nb if nb<=8 else (12 if nb<=12 else 16 if nb<=16 else 24 if nb<=24 else 32 if nb<=32 else 64)
    bynb[k][0]+=c; bynb[k][1]+=c*nb*mid(mb); bynb[k][2]+=c*(mid(mb)/16+nb)
acc=[0,0,0]
for k in sorted(bynb):
    v=bynb[k]; acc=[a+b for a,b in zip(acc,v)]
    print("nb<=%2d: problems %5.1f%% (cum %5.1f%%)  block steps %5.1f%% (cum %5.1f%%)  tiles %5.1f%% (cum %5.1f%%)"%(k,100*v[0]/tot,100*acc[0]/tot,100*v[1]/steps,100*acc[1]/steps,100*v[2]/tiles,100*acc[2]/tiles))
bym=collections.defaultdict(lambda:[0,0])
for (mo,nb,mb),c in cnt.items():
    if nb<=2: bym[mb][0]+=c; bym[mb][1]+=c*nb*mid(mb)
print("nb<=2 by target bucket (32 columns each):", {mb:v[0] for mb,v in sorted(bym.items())})
shw=sum(c for (mo,nb,mb),c in cnt.items() if mo==1); print("SHW problems",shw)
PY
cat $OUT/hist.txt; tail -3 $OUT/b.err
