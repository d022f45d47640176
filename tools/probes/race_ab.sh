# steady-state in-flight stress (3 slots) under a few configurations: wrong results per N replays
cd $GRAFT_REPO_ROOT
export PROBE_SLOTS=3 PROBE_ROUNDS=${1:-400}
run() { echo "== $1"; env $2 python tools/probes/inflight_race.py 2>&1 | grep -v "eager vs\|amdgpu.ids" | tail -n 3; }
run "default" "X=1"
run "separate head kernels" "CFP_HEAD_FUSED=0"
run "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8"
cd cfpnet_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-variable -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DHF_RAW_BARRIER -c head_fused.hip -o head_fused.o
make 2>&1 | grep -E "error" || true
cd ../..
run "fused head with raw s_barrier after vmcnt(0) only (the form that failed)" "X=1"
