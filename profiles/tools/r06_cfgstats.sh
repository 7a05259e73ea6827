# rocprofv3 --kernel-trace --stats of one C4 / C5 step in the exclusive pass's mode (one lane, one chunk at a time, queues one after the other): per-kernel totals
mkdir -p gpurun_out/r06_configs
R=$PWD
cd /tmp && export TMPDIR=/tmp
export LF_LANES=1 LF_SERIAL_CLASSES=1 LF_CHUNK_BASES=1099511627776
for cfg in c4 c5; do
if [ $cfg = c4 ]; then export LF_CHUNK_READS=25000; else export LF_CHUNK_READS=1073741824; fi
rm -rf /tmp/lfp_cfg_$cfg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lfp_cfg_$cfg -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive --no-host-region > $R/gpurun_out/r06_configs/bench_${cfg}_serialized_under_rocprof.json 2> /tmp/lfp_cfg_$cfg.err
python3 $R/profiles/tools/trim_stats.py $(ls /tmp/lfp_cfg_$cfg/*/*kernel_stats.csv | head -1) $R/gpurun_out/r06_configs/kernel_stats_serialized_$cfg.csv
head -8 $R/gpurun_out/r06_configs/kernel_stats_serialized_$cfg.csv | cut -c1-160
done
