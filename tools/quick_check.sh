# forward parity tests + a bench line without the CPU baseline / training legs; outputs under gpurun_out/$1
#   bash tools/quick_check.sh r3w
TAG=${1:-quick}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_forward_gpu.py tests/test_ops_gpu.py -q -x -k "not stream_kernel" > gpurun_out/$TAG/fwd_tests.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/$TAG/fwd_tests.log
timeout 500 python bench.py --no-cpu-baseline --no-train > gpurun_out/$TAG/bench_quick.json 2> gpurun_out/$TAG/bench_quick.err; echo "bench rc=$?"
python - "$TAG" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}/bench_quick.json").read().strip().splitlines()[-1])
print("maps/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 4), "rel_l1", d.get("rel_l1"), "f16", d.get("f16", {}).get("value"), d.get("f16", {}).get("rel_l1"))
print("latency", d.get("latency"), d["config"].get("lane_choice_ms_per_step"))
print("roofline frac", d["roofline"]["frac"], "dw in_graph", d["dw3x3"].get("in_graph"), d["dw3x3"].get("frac_of_measured_copy_in_graph"))
print("kernel_ms_per_step", d.get("kernel_ms_per_step"))
PY
