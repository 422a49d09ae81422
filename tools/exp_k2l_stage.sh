export TMPDIR=/tmp; mkdir -p gpurun_out/r02c
for st in 0 128 256 384; do
MP_K2L_STAGE=$st rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/stats_st$st -o run -- python3 bench.py --config C --steps 5 --warmup 1 --cpu-sample 0 --no-consume > gpurun_out/r02c/stats_st$st.log 2>&1
echo "stage $st"; grep "k2l" gpurun_out/r02c/stats_st$st/run_kernel_stats.csv | cut -d, -f1,4
done
