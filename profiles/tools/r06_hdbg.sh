# LF_HIRSCH_DEBUG runs of C4 and C5 (T2T-like): nodes per class and level, trial bounds, roots by distance / rows
# (no LF_WATCHDOG in timing runs)
mkdir -p gpurun_out/r6_hdbg
LF_HIRSCH_DEBUG=1 timeout 900 python3 bench.py --config c4 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region > gpurun_out/r6_hdbg/c4.json 2> gpurun_out/r6_hdbg/c4_levels.txt
LF_HIRSCH_DEBUG=1 timeout 1200 python3 bench.py --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive --no-host-region > gpurun_out/r6_hdbg/c5.json 2> gpurun_out/r6_hdbg/c5_levels.txt
grep "roots by" gpurun_out/r6_hdbg/c4_levels.txt | head -12
grep "roots by" gpurun_out/r6_hdbg/c5_levels.txt | head -12
