#!/bin/bash
# SQ instruction / cycle counters and HBM traffic counters of one bench configuration, each in its own --pmc pass
# (the program itself after `--`; no trace domains combined with --pmc). Output: gpurun_out/<tag>/...
#   tools/profile_sq.sh <tag> <config> [extra bench args]
set -e -o pipefail
TAG=$1; CFG=$2; shift 2
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
B="bench.py --config $CFG --steps 2 --warmup 1 --cpu-sample 0 --no-consume $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$CFG -o run -- python3 bench.py --config $CFG --steps 5 --warmup 1 --cpu-sample 0 --no-consume $* > $O/stats_$CFG.log 2>&1
echo "stats $CFG done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq1_$CFG -o run -- python3 $B > $O/sq1_$CFG.log 2>&1
echo "sq1 $CFG done"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT --output-format csv -d $O/sq2_$CFG -o run -- python3 $B > $O/sq2_$CFG.log 2>&1
echo "sq2 $CFG done"
# (a pass with TA_* / TCP_* counters hung on this pool - "incomplete dispatches", killed after 7 minutes of silence: not collected)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$CFG -o run -- python3 $B > $O/fetch_$CFG.log 2>&1
echo "fetch $CFG done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$CFG -o run -- python3 $B > $O/write_$CFG.log 2>&1
echo "write $CFG done"
F=$(find $O -name "*counter_collection.csv" | grep "_$CFG/" | sort)
python3 tools/pmc_summary.py $F > $O/pmc_$CFG.json
S=$(find $O/stats_$CFG -name "*kernel_stats.csv" | head -1)
cp $S $O/kernel_stats_$CFG.csv
echo "summary $CFG done"
