// Does the f32 GEMM epilogue's store pattern cost bandwidth?  A wave of the f16x3 kernels stores, per instruction, 16 rows x 64 contiguous bytes
// (lane = (row fr, column quad fq)); the alternative after an in-wave transpose is 4 rows x 256 contiguous bytes.  Both write the same
// [M][N] float32 matrix, 64 x 64 tiles per 256-thread workgroup, nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/store_pattern tools/probes/store_pattern.hip && gpurun_out/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ out, int M, int N) {
  const int tiles_n = (N + 63) / 64;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = tile_m * 64 + wave * 16, n0 = tile_n * 64;
  const f32x4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  if (MODE == 0) {            // lane = (row fr, quad fq): per instruction 16 rows x 64 B
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + fr, n = n0 + j * 16 + fq * 4;
      if (m < M && n < N) *reinterpret_cast<f32x4*>(out + (long long)m * N + n) = v;
    }
  } else {                    // lane = (row r4 of 4, quad c of 16): per instruction 4 rows x 256 B
    const int r4 = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = m0 + t * 4 + r4, n = n0 + c * 4;
      if (m < M && n < N) *reinterpret_cast<f32x4*>(out + (long long)m * N + n) = v;
    }
  }
}

int main() {
  const int shapes[][2] = {{9600, 816}, {2400, 1392}, {614400, 256}, {153600, 96}, {38400, 224}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1];
    float* out;
    hipMalloc(&out, (size_t)M * N * 4);
    const int grid = ((M + 63) / 64) * ((N + 63) / 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep)
      for (int mode = 0; mode < 2; ++mode) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) {
          if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(256), 0, 0, out, M, N);
          else hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(256), 0, 0, out, M, N);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best[mode]) best[mode] = ms / 20;
      }
    printf("%7d x %5d f32 (%6.1f MB): 16 rows x 64 B per store %7.1f us (%5.2f TB/s)   4 rows x 256 B %7.1f us (%5.2f TB/s)\n", M, N, M * (double)N * 4 / 1e6,
           best[0] * 1e3, M * (double)N * 4 / best[0] / 1e9, best[1] * 1e3, M * (double)N * 4 / best[1] / 1e9);
    hipFree(out);
  }
  return 0;
}
