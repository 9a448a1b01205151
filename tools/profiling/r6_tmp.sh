cd /root/repo
for n in 32 16 8 4; do NENV=$n python tools/profiling/raster_bench.py sloth_32env 2>/dev/null | tail -1; done
