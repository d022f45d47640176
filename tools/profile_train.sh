# kernel-trace profile of the replayed training step (16 crops of 416x544, one HIP graph); summary under gpurun_out/
#   bash tools/profile_train.sh r3 f16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}
DT=${2:-f16}
RAW=/tmp/proft_$TAG
mkdir -p $RAW
cd $R
rocprofv3 --kernel-trace --stats -d $RAW/kt -o kt -- python3 tools/train_bench.py --dtype $DT --graph --steps 8 > gpurun_out/${TAG}_train_bench_${DT}.json 2> $RAW/kt.err
DB=$(ls $RAW/kt/*.db | head -n 1)
python3 tools/train_profile_summary.py $DB 12 16 > gpurun_out/${TAG}_train_step_b16_${DT}_kernel_stats.csv
head -n 40 gpurun_out/${TAG}_train_step_b16_${DT}_kernel_stats.csv
tail -n 2 gpurun_out/${TAG}_train_bench_${DT}.json
