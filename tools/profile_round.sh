#!/bin/bash
# Round-end measurements on the GPU box; everything lands in gpurun_out/ (copy the summaries into profiles/ afterwards).
set -e -o pipefail
R=${1:-r01}
O=gpurun_out/$R
mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 > $O/bench_configC.json 2> $O/bench_configC.err
echo "bench C done"; tail -c 600 $O/bench_configC.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-consume > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-consume > $O/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-consume > $O/pmc_write.log 2>&1
echo "write done"
python bench.py --config B --steps 10 --warmup 2 --cpu-sample 0 > $O/bench_configB.json 2> $O/bench_configB.err
python bench.py --config D --steps 5 --warmup 1 --cpu-sample 0 > $O/bench_configD.json 2> $O/bench_configD.err
echo "B, D done"
find $O -name "*.csv" | head -20
