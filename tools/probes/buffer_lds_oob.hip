// Probe: what does `buffer_load_dwordx4 ... offen lds` write to LDS for a lane whose offset fails the descriptor's range check?
// (zeros / nothing).  Also: is soffset part of the range check?   hipcc --offload-arch=gfx950 -O3 -o /tmp/oob tools/probes/buffer_lds_oob.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const unsigned short* in, unsigned short* out, int nbytes, int soff, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[1024];
  const int lane = threadIdx.x;
  for (int i = 0; i < 8; ++i) ((unsigned short*)sm)[lane * 8 + i] = 0xBEEF;      // sentinel
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
  int voff = lane * 16;
  if (mode == 1 && (lane & 1)) voff = 0x7ffffff0;          // far out of range
  if (mode == 2 && (lane & 1)) voff = nbytes;              // first byte past the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)sm, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 8; ++i) out[lane * 8 + i] = ((unsigned short*)sm)[lane * 8 + i];
}
int main() {
  const int n = 64 * 8 * 4;
  std::vector<unsigned short> h(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i + 1);
  unsigned short *din, *dout;
  hipMalloc(&din, n * 2); hipMalloc(&dout, 64 * 8 * 2);
  hipMemcpy(din, h.data(), n * 2, hipMemcpyHostToDevice);
  std::vector<unsigned short> o(64 * 8);
  struct { int nbytes, soff, mode; const char* what; } cases[] = {
    {n * 2, 0, 0, "all lanes in range"}, {n * 2, 0, 1, "odd lanes voffset 0x7ffffff0"}, {64 * 16, 0, 2, "odd lanes voffset == num_records"},
    {64 * 16, 1024, 0, "soffset 1024 pushes every lane past num_records (voffset itself in range)"},
    {n * 2, 1024, 0, "soffset 1024, everything in range"}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, c.nbytes, c.soff, c.mode);
    hipMemcpy(o.data(), dout, 64 * 8 * 2, hipMemcpyDeviceToHost);
    printf("%-80s lane0: %04x %04x  lane1: %04x %04x  lane2: %04x  lane3: %04x\n", c.what, o[0], o[1], o[8], o[9], o[16], o[24]);
  }
  return 0;
}
