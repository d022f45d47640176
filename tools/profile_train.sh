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
# HBM traffic of the same step (FETCH_SIZE / WRITE_SIZE in their own passes, as tools/profile_round.sh): per kernel family, bytes per launch
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/pf -o f --output-format csv -- python3 tools/train_bench.py --dtype $DT --graph --steps 3 > /dev/null 2> $RAW/pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/pw -o w --output-format csv -- python3 tools/train_bench.py --dtype $DT --graph --steps 3 > /dev/null 2> $RAW/pw.err
python3 tools/pmc_traffic.py $RAW/pf/f_counter_collection.csv $RAW/pw/w_counter_collection.csv > gpurun_out/${TAG}_train_pmc.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_train_pmc.json"))
rows = sorted(((v["hbm_bytes_per_launch"] * v["launches_fetch_pass"], k, v) for k, v in d.items() if isinstance(v, dict)), reverse=True)
tot = sum(r[0] for r in rows)
print(f"HBM bytes over the pass: {tot / 1e9:.1f} GB; largest families:")
for b, k, v in rows[:14]:
    print(f"  {k[:60]:60s} {v['launches_fetch_pass']:6d} launches  {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB each  {100 * b / tot:5.1f} %")
PY
