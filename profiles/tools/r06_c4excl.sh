# the Hirschberg levels of C4's exclusive pass (100 k reads as one chunk, classes one after the other), banded and unbanded
mkdir -p gpurun_out/r6_c4x
for band in 1 0; do
LF_HIRSCH_BAND=$band LF_HIRSCH_DEBUG=1 timeout 900 python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-host-region > gpurun_out/r6_c4x/c4_band$band.json 2> gpurun_out/r6_c4x/c4_levels_band$band.txt
echo "== band $band"; grep -n "hirschberg: " gpurun_out/r6_c4x/c4_levels_band$band.txt | tail -8
done
