// TEST TOOL: what one dependent LDS round trip costs a lone wavefront on gfx950 (the pack engines are one wave per problem)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) chain_b32(unsigned* out, int iters, int lds_words) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < lds_words; i += 64) lds[i] = (unsigned)((i * 7 + 13) % lds_words);
  __syncthreads();
  unsigned x = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) x = lds[x];
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { out[64] = (unsigned)((t1 - t0) / iters); }
}
__global__ void __launch_bounds__(64) chain3(unsigned* out, int iters, int n) {
  // u16 order -> 24-byte record -> 32-byte record, like the cursor engine's scan; one ballot per step
  extern __shared__ unsigned lds[];
  unsigned short* ord = (unsigned short*)lds;                 // n
  unsigned long long* rec = (unsigned long long*)(lds + 4096);  // n * 3
  unsigned long long* ent = (unsigned long long*)(lds + 4096 + 6 * 4096);  // 1024 * 4
  for (int i = threadIdx.x; i < n; i += 64) { ord[i] = (unsigned short)((i * 5 + 3) % n); rec[i * 3] = i * 0x9E3779B97F4A7C15ull; rec[i * 3 + 1] = i; rec[i * 3 + 2] = 2 * i; }
  for (int i = threadIdx.x; i < 4096; i += 64) ent[i] = i;
  __syncthreads();
  unsigned r = 0;
  unsigned long long acc = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    unsigned p = (r + threadIdx.x) % n;
    unsigned x = ord[p];
    unsigned long long a = rec[x * 3], b = rec[x * 3 + 1], c = rec[x * 3 + 2];
    unsigned h = (unsigned)((a * 0x9E3779B97F4A7C15ull) >> 40) & 1023;
    unsigned long long e0 = ent[h * 4], e1 = ent[h * 4 + 1], e2 = ent[h * 4 + 2], e3 = ent[h * 4 + 3];
    unsigned long long m = __ballot((e0 + e1 + e2 + e3 + b + c) & 1);
    acc += m;
    r = (unsigned)__builtin_ctzll(m | (1ull << 63)) + r + 1;
    if (r >= (unsigned)n) r -= n;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = (unsigned)acc;
  if (threadIdx.x == 0) out[64] = (unsigned)((t1 - t0) / iters);
}
int main() {
  unsigned* d; hipMalloc(&d, 65 * 4);
  unsigned h[65];
  for (int lds_kb : {16, 64, 150}) {
    hipFuncSetAttribute((const void*)chain_b32, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipLaunchKernelGGL(chain_b32, dim3(1), dim3(64), lds_kb * 1024, 0, d, 200000, lds_kb * 256);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("dependent ds_read_b32 chain, %d KB LDS: %u cycles per trip\n", lds_kb, h[64]);
  }
  hipFuncSetAttribute((const void*)chain3, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(chain3, dim3(1), dim3(64), 150 * 1024, 0, d, 200000, 2763);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("3-stage dependent scan step (u16 -> 24 B -> 32 B, ballot): %u cycles per step\n", h[64]);
  return 0;
}
