#!/bin/bash
# Kernel times and counters of BASELINE config E (normal -> build_reference -> somatic -> filter, tools/config_e_run.py) on one GPU:
# one --kernel-trace --stats run, then the SQ / FETCH_SIZE / WRITE_SIZE counters each in its own --pmc pass (the program itself
# after `--`; no trace domain combined with --pmc). Output: gpurun_out/<tag>/...
#   tools/profile_config_e.sh <tag> [transcripts]
set -e -o pipefail
TAG=$1; N=${2:-20000}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
R="tools/config_e_run.py --transcripts $N"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_E -o run -- python3 $R --out $O/config_e_run.json > $O/stats_E.log 2>&1
echo "stats E done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq1_E -o run -- python3 $R > $O/sq1_E.log 2>&1
echo "sq1 E done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_E -o run -- python3 $R > $O/fetch_E.log 2>&1
echo "fetch E done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_E -o run -- python3 $R > $O/write_E.log 2>&1
echo "write E done"
F=$(find $O -name "*counter_collection.csv" | grep "_E/" | sort)
python3 tools/pmc_summary.py --sum $F > $O/pmc_configE.json
S=$(find $O/stats_E -name "*kernel_stats.csv" | head -1)
cp $S $O/kernel_stats_configE.csv
echo "summary E done"
