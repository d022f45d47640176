# Everything the round's documentation quotes, in one GPU call; outputs under gpurun_out/$1 (copy what is to be judged to profiles/)
#   bash tools/final_round.sh r3c
TAG=${1:-final}
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/bench_b8.json 2> $O/bench_b8.err; echo "bench rc=$?"
python bench.py --config5 > $O/config5.json 2> $O/config5.err; echo "config5 rc=$?"
python bench.py --train --steps 30 --warmup 5 > $O/train_line.json 2> $O/train_line.err; echo "train rc=$?"
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1; echo "profile rc=$?"
bash tools/profile_train.sh $TAG f16 > $O/profile_train.log 2>&1; echo "profile train rc=$?"
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads(open(f"{o}/bench_b8.json").read().strip().splitlines()[-1])
print("maps/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), d.get("rel_l1"), "f16", d.get("f16", {}).get("value"), d.get("f16", {}).get("rel_l1"))
print("latency", d.get("latency"))
print("roofline", d["roofline"]["frac"], "dw", d["dw3x3"].get("frac_of_measured_copy_in_graph"), d["dw3x3"].get("frac_of_measured_copy"))
print("training", {k: v for k, v in d.get("training", {}).items() if k in ("value", "ms_per_step", "dtype")})
PY
