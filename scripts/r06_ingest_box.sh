#!/bin/bash
# the ingest's decoder on the GPU box's host cores: the bare decoder against zlib (one thread, 16 threads), then the gzip -6 file of
# scripts/ingest_gz_bench.py with the trace of its stages.  Host only.   bash scripts/r06_ingest_box.sh > gpurun_out/...
set -e
grep -m1 "model name" /proc/cpuinfo; echo "hardware threads visible: $(nproc); transparent huge pages: $(cat /sys/kernel/mm/transparent_hugepage/enabled), defrag $(cat /sys/kernel/mm/transparent_hugepage/defrag)"
g++ -O3 -std=c++17 -o /tmp/inflate_rate scripts/micro/inflate_rate.cpp -lz -lpthread
/tmp/inflate_rate 100 16 | grep "16 thread"
DEBWT_GZBENCH_ONLY=x.gzip6,x.bgzf DEBWT_GZBENCH_KEEP=1 python scripts/ingest_gz_bench.py 1000
for f in x.gzip6.fa.gz x.bgzf.fa.gz; do
  DEBWT_TRACE_GZ=1 python -c "
from debwt_amd import api
import time
for i in range(2):
    t = time.time(); r = api.pack_fasta('/dev/shm/debwt_gzbench/$f', 16); print('$f', round(time.time() - t, 3), 'read/inflate', round(r[3], 3), 'pack', round(r[4], 3))
" 2>&1 | grep -v "gz_parallel: piece"
done
rm -rf /dev/shm/debwt_gzbench
