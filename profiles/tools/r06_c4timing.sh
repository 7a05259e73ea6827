# one HBM-resident C4 step with LF_TIMING=1, banded and unbanded Hirschberg levels: what a 25 k-read chunk's time is made of
# (no LF_WATCHDOG in timing runs)
mkdir -p gpurun_out/r6_c4t
for band in 1 0 1 0; do
LF_HIRSCH_BAND=$band LF_TIMING=1 timeout 900 python3 bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > gpurun_out/r6_c4t/c4_band$band.json 2> gpurun_out/r6_c4t/c4_timing_band$band.txt
echo "== band $band"
grep -E "timeline|lf_map_batch total|merge\+solve" gpurun_out/r6_c4t/c4_timing_band$band.txt | tail -22
done
