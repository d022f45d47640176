# profiles of one round on the GPU box: kernel-trace stats, two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run), SQ counters.
# The raw traces are tens of MB: they are summarised here and only the summaries stay under gpurun_out/ (copy them to profiles/).
#   bash tools/profile_round.sh r2x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
RAW=/tmp/prof_$TAG
mkdir -p $RAW
cd $R
BENCH="python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --no-x3"
rocprofv3 --kernel-trace --stats -d $RAW/kt -o kt --output-format csv -- $BENCH > gpurun_out/${TAG}_prof_bench.json 2> $RAW/kt.err
# forwards in that trace: 2 warm-up + 1 capture pass per candidate slot (lanes:1, inflight 2/3/4 = 1+2+3+4 slots) are eager, everything else replays
python3 tools/prof_summary_csv.py $RAW/kt/kt_kernel_trace.csv --csv gpurun_out/${TAG}_bench_b8_kernel_stats.csv > gpurun_out/${TAG}_bench_b8_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/pf -o f --output-format csv -- $BENCH --steps 3 --warmup 1 > /dev/null 2> $RAW/pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/pw -o w --output-format csv -- $BENCH --steps 3 --warmup 1 > /dev/null 2> $RAW/pw.err
python3 tools/pmc_traffic.py $RAW/pf/f_counter_collection.csv $RAW/pw/w_counter_collection.csv > gpurun_out/${TAG}_pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $RAW/ps -o s --output-format csv -- $BENCH --steps 3 --warmup 1 > /dev/null 2> $RAW/ps.err
python3 tools/pmc_sq_summary.py $RAW/ps/s_counter_collection.csv > gpurun_out/${TAG}_pmc_sq_inference.json
head -n 30 gpurun_out/${TAG}_bench_b8_kernel_stats.txt
# the default boundary mode (float32 storage, f16x3 matrix math): kernel trace, HBM traffic and SQ counters of tools/mode_bench.py in that mode
export MODE_BENCH_DTYPES=x3 MODE_BENCH_MAX_INFLIGHT=4
X3="python3 tools/mode_bench.py 8"
rocprofv3 --kernel-trace -d $RAW/x3kt -o kt --output-format csv -- $X3 > gpurun_out/${TAG}_x3_mode.json 2> $RAW/x3kt.err
python3 tools/prof_summary_csv.py $RAW/x3kt/kt_kernel_trace.csv --csv gpurun_out/${TAG}_x3_mode_kernel_stats.csv > gpurun_out/${TAG}_x3_mode_kernel_stats.txt
export MODE_BENCH_MAX_INFLIGHT=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/x3pf -o f --output-format csv -- $X3 > /dev/null 2> $RAW/x3pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/x3pw -o w --output-format csv -- $X3 > /dev/null 2> $RAW/x3pw.err
python3 tools/pmc_traffic.py $RAW/x3pf/f_counter_collection.csv $RAW/x3pw/w_counter_collection.csv > gpurun_out/${TAG}_x3_pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $RAW/x3ps -o s --output-format csv -- $X3 > /dev/null 2> $RAW/x3ps.err
python3 tools/pmc_sq_summary.py $RAW/x3ps/s_counter_collection.csv > gpurun_out/${TAG}_x3_pmc_sq.json
head -n 24 gpurun_out/${TAG}_x3_mode_kernel_stats.txt
