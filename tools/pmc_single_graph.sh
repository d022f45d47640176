# PMC traffic of the 16-bit forward with ONE graph in flight (--lanes 1) next to the default (best of 1-4 in flight): does the fetch beyond L2 per
# implicit-GEMM launch depend on how many forwards share the L2s?   bash tools/pmc_single_graph.sh r4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r4}
RAW=/tmp/pmc1_$TAG
mkdir -p $RAW
cd $R
B1="python3 bench.py --no-cpu-baseline --no-kernel-times --no-f16 --no-train --no-x3 --lanes 1 --steps 12 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/pf -o f --output-format csv -- $B1 > /dev/null 2> $RAW/pf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/pw -o w --output-format csv -- $B1 > /dev/null 2> $RAW/pw.err
python3 tools/pmc_traffic.py $RAW/pf/f_counter_collection.csv $RAW/pw/w_counter_collection.csv > gpurun_out/${TAG}_pmc_traffic_single_graph.json
python3 - <<PY
import json
a = json.load(open("gpurun_out/${TAG}_pmc_traffic_single_graph.json"))
import glob
b = json.load(open(sorted(glob.glob("profiles/r4z_pmc_traffic.json"))[-1]))
print(f"{'kernel family':44s} {'1 graph: FETCHx2 KiB':>20s} {'WRITE KiB':>10s} | {'in flight: FETCHx2':>18s} {'WRITE':>10s}")
for k in sorted(a):
    if k in b and ("igemm2" in k or "halo" in k or "dw3x3" in k or "head" in k):
        print(f"{k:44s} {2 * a[k]['FETCH_SIZE_KiB_avg']:20.0f} {a[k]['WRITE_SIZE_KiB_avg']:10.0f} | {2 * b[k]['FETCH_SIZE_KiB_avg']:18.0f} {b[k]['WRITE_SIZE_KiB_avg']:10.0f}")
PY
