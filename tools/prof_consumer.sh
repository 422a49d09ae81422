#!/bin/bash
# Development aid: single-thread run of a -DMP_PROFILE build of the CLI on a synthetic exome (section cycle counts on stderr).
#   tools/prof_consumer.sh <binary> [transcripts]
# The binary: the CLI linked from the library's objects with consume.cpp compiled -DMP_PROFILE, e.g. in microphaser_amd/csrc:
#   hipcc --offload-arch=gfx950 -std=c++17 -O3 -DMP_PROFILE -c consume.cpp -o /tmp/consume_prof.o && hipcc -O3 -std=c++17 -c cli.cpp -o /tmp/cli.o &&
#   hipcc --offload-arch=gfx950 -o ../_lib/microphaser_prof /tmp/cli.o /tmp/consume_prof.o $(ls ../_lib/*.o | grep -v consume.o) -lz -lpthread
set -e
BIN=${1:-microphaser_amd/_lib/microphaser_prof}
N=${2:-2000}
python -c "
import sys; sys.path.insert(0, '.')
import microphaser_amd as m
m.Context(-1).synth(2020, $N, 30.0, 5.4).write('/tmp/cb')
"
MP_CLEAN_EXIT=1 MP_THREADS=1 MP_DEBUG=1 timeout -k 10 300 $BIN somatic /tmp/cb.bam --variants /tmp/cb.vcf --ref /tmp/cb.fa --tsv /tmp/o.tsv --normal-output /tmp/o.n.fa < /tmp/cb.gtf 2> /tmp/err.log > /tmp/o.fa
grep -E "^\[prof\]|batch_results|batch_create|consume" /tmp/err.log | cut -c1-140
md5sum /tmp/o.tsv /tmp/o.fa /tmp/o.n.fa
