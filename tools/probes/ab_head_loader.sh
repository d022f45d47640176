# A/B of the fused head's main-loop loader on the GPU box: descriptors (as shipped) vs global_load_lds + pointer select (-DHF_GLDS)
set -e
cd $GRAFT_REPO_ROOT
python tools/head_bench.py 2>&1 | grep -E "one kernel, Wout hi\+lo=False|GEMM1 only  " | head -4
cd cfpnet_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-variable -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DHF_GLDS -c head_fused.hip -o head_fused.o
make 2>&1 | grep -E "error" || true
cd ../..
echo "---- -DHF_GLDS"
python -m pytest tests/test_ops_gpu.py -q -m gpu -k depth_head_fused 2>&1 | tail -n 1
python tools/head_bench.py 2>&1 | grep -E "one kernel, Wout hi\+lo=False|GEMM1 only  " | head -4
