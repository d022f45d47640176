# A/B of the two gen-2 GEMM loaders on the GPU box (first the tree as it is = global_load_lds + pointer select, then the descriptor variant)
set -e
cd $GRAFT_REPO_ROOT
python tools/mode_bench.py 2>&1 | grep -v float32 | tail -n 2
python tools/big_gemm_bench.py 2>&1 | tail -n 6
cp tools/probes/conv_igemm2_buffer_loader.hip.txt cfpnet_amd/csrc/conv_igemm2.hip
make -C cfpnet_amd/csrc -j16 2>&1 | grep -E "error" || true
echo "---- buffer-descriptor loader (buffer_load ... lds, 32-bit offsets, hardware zero fill, scalar tap offsets)"
python tools/mode_bench.py 2>&1 | grep -v float32 | tail -n 2
python tools/big_gemm_bench.py 2>&1 | tail -n 6
