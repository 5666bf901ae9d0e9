#!/bin/bash
# the ingest on the GPU box's host cores (no GPU is used): the bare DEFLATE decoder and CRC-32 against zlib's (scripts/micro/
# inflate_rate.cpp), the table of scripts/ingest_gz_bench.py, and the stage trace of the one-member and the block-gzip file.
#   bash scripts/r06_ingest_box.sh > gpurun_out/r06_ingest_gz_rate.txt
set -e
grep -m1 "model name" /proc/cpuinfo; echo "hardware threads visible: $(nproc); transparent huge pages: $(cat /sys/kernel/mm/transparent_hugepage/enabled), defrag $(cat /sys/kernel/mm/transparent_hugepage/defrag)"
g++ -O3 -std=c++17 -o /tmp/inflate_rate scripts/micro/inflate_rate.cpp -lz -lpthread
/tmp/inflate_rate 100 16
DEBWT_GZBENCH_KEEP=1 python scripts/ingest_gz_bench.py 1000
for f in x.gzip6.fa.gz x.gzip1.fa.gz x.members24.fa.gz x.members3.fa.gz; do
  DEBWT_TRACE_GZ=1 python -c "
from debwt_amd import api
import time
for i in range(2):
    time.sleep(0.5); t = time.time(); r = api.pack_fasta('/dev/shm/debwt_gzbench/$f', 16); print('$f', 'read/inflate', round(r[3], 3), 'pack', round(r[4], 3))
" 2>&1 | grep -v "gz_parallel: piece" | tail -7
done
rm -rf /dev/shm/debwt_gzbench
echo "## what filling and releasing 1 GB costs on this host (scripts/micro/page_cost.cpp)"
g++ -O2 -std=c++17 -o /tmp/page_cost scripts/micro/page_cost.cpp -lpthread && /tmp/page_cost
