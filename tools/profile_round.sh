# profiles of one round: kernel-trace stats, two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run), a full bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1t -o r1t -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train > gpurun_out/prof_r1t_bench.json 2> gpurun_out/prof_r1t.err
ls -R gpurun_out/prof_r1t | head -20
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_write.err
ls -R gpurun_out/pmc_fetch gpurun_out/pmc_write | head
python bench.py > gpurun_out/r1t_bench_b8.json 2> gpurun_out/r1t_bench.err
tail -c 600 gpurun_out/r1t_bench_b8.json
