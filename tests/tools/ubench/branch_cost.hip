// TEST TOOL: what a taken / not-taken scalar branch costs a lone wavefront on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
__global__ void __launch_bounds__(64) k_taken(unsigned* out, int iters) {
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("s_branch 1f\n\ts_nop 0\n\ts_nop 0\n1:\n\tv_mov_b32 v1, v1" ::: "v1");)
    REP8(asm volatile("s_branch 1f\n\ts_nop 0\n\ts_nop 0\n1:\n\tv_mov_b32 v1, v1" ::: "v1");)
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (unsigned)((t1 - t0) * 100 / iters / 16);
}
__global__ void __launch_bounds__(64) k_nottaken(unsigned* out, int iters) {
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("s_cmp_eq_u32 s0, s0\n\ts_cbranch_scc0 1f\n\ts_nop 0\n1:\n\tv_mov_b32 v1, v1" ::: "v1", "scc");)
    REP8(asm volatile("s_cmp_eq_u32 s0, s0\n\ts_cbranch_scc0 1f\n\ts_nop 0\n1:\n\tv_mov_b32 v1, v1" ::: "v1", "scc");)
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (unsigned)((t1 - t0) * 100 / iters / 16);
}
__global__ void __launch_bounds__(64) k_plain(unsigned* out, int iters) {
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("v_mov_b32 v1, v1\n\ts_nop 0\n\ts_add_u32 s0, s0, 0" ::: "v1", "scc");)
    REP8(asm volatile("v_mov_b32 v1, v1\n\ts_nop 0\n\ts_add_u32 s0, s0, 0" ::: "v1", "scc");)
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (unsigned)((t1 - t0) * 100 / iters / 16);
}
__global__ void __launch_bounds__(64) k_execbr(unsigned* out, int iters) {
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    REP8(asm volatile("s_and_saveexec_b64 s[2:3], vcc\n\ts_cbranch_execz 1f\n\tv_mov_b32 v1, v1\n1:\n\ts_or_b64 exec, exec, s[2:3]" ::: "v1", "s2", "s3", "scc");)
    REP8(asm volatile("s_and_saveexec_b64 s[2:3], vcc\n\ts_cbranch_execz 1f\n\tv_mov_b32 v1, v1\n1:\n\ts_or_b64 exec, exec, s[2:3]" ::: "v1", "s2", "s3", "scc");)
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = (unsigned)((t1 - t0) * 100 / iters / 16);
}
int main() {
  unsigned* d; hipMalloc(&d, 4); unsigned h;
  hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, 0, d, 100000); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("3 plain instructions (v_mov, s_nop, s_add): %.2f cycles per group\n", h / 100.0);
  hipLaunchKernelGGL(k_taken, dim3(1), dim3(64), 0, 0, d, 100000); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("taken s_branch + v_mov: %.2f cycles per group\n", h / 100.0);
  hipLaunchKernelGGL(k_nottaken, dim3(1), dim3(64), 0, 0, d, 100000); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("s_cmp + not-taken s_cbranch + s_nop + v_mov: %.2f cycles per group\n", h / 100.0);
  hipLaunchKernelGGL(k_execbr, dim3(1), dim3(64), 0, 0, d, 100000); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("saveexec + execz branch (not taken) + v_mov + restore: %.2f cycles per group\n", h / 100.0);
  return 0;
}
