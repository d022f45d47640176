cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export MODE_BENCH_DTYPES=x3 MODE_BENCH_MAX_INFLIGHT=1
rocprofv3 --kernel-trace -d /tmp/latx -o kt --output-format csv -- python3 tools/mode_bench.py 1 > gpurun_out/lat_x3_b1.json 2> /tmp/latx.err
python3 tools/prof_summary_csv.py /tmp/latx/kt_kernel_trace.csv --csv gpurun_out/lat_x3_b1_kernel_stats.csv > gpurun_out/lat_x3_b1_kernel_stats.txt
head -45 gpurun_out/lat_x3_b1_kernel_stats.txt
