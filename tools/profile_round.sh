# profiles of one round: kernel-trace stats, two PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run), SQ counters, a full bench line
# usage on the GPU box:  bash tools/profile_round.sh r2x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_$TAG.err
ls -R gpurun_out/prof_$TAG | head -20
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch_$TAG -o f --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_fetch_$TAG.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write_$TAG -o w --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_write_$TAG.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/pmc_sq_$TAG -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_sq_$TAG.err
find gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG -name "*.csv" | head -20
