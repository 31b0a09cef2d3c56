// ksolve_pack_batch.hip — the general engine on many problems per launch: batches and the one-wavefront consolidation sweep.
#include "pack_kernels.h"

// (round 6) The view and the workspace record a block works on are copied into LDS (ks::kSweepLdsExtra bytes behind the plan's own)
// before an engine is built on them: the engine reads their fields all through a solve, and from HBM every such read was a vector
// load with an L2 round trip in front of its use (ksolve_pack_sweep4.hip has the measurement: 1.80 -> 1.44 ms per 10k probes).
static __device__ __forceinline__ void ks_copy_words(void* dst, const void* src, int bytes) {
  const uint64_t* s = (const uint64_t*)src;
  uint64_t* d = (uint64_t*)dst;
  for (int i = (int)(threadIdx.x & 63); i < bytes / 8; i += 64) d[i] = s[i];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
constexpr int kViewBytes = (int)((sizeof(ks::ProblemView) + 15) & ~(size_t)15);
static_assert(kViewBytes + (int)sizeof(ks::Workspace) <= ks::kSweepLdsExtra, "kSweepLdsExtra");

// Batched form: block b solves problem b (its view and workspace are read from HBM instead of the kernel arguments).
__global__ void __launch_bounds__(64) ksolve_pack_batch(ks::BatchItem* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::BatchItem& it = items[blockIdx.x];
  char* const extra = lds + ((it.pv.lds.total_bytes + 15) & ~15);
  ks_copy_words(extra, &it.pv, (int)sizeof(ks::ProblemView));
  ks_copy_words(extra + kViewBytes, &it.ws, (int)sizeof(ks::Workspace));
  const ks::ProblemView& pv = *(const ks::ProblemView*)extra;
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  ks::Engine<ks::Wave, true> eng(pv, *(ks::Workspace*)(extra + kViewBytes), tables);
  eng.solve();
}
__global__ void __launch_bounds__(64) ksolve_pack_batch_lite(ks::BatchItem* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::BatchItem& it = items[blockIdx.x];
  char* const extra = lds + ((it.pv.lds.total_bytes + 15) & ~15);
  ks_copy_words(extra, &it.pv, (int)sizeof(ks::ProblemView));
  ks_copy_words(extra + kViewBytes, &it.ws, (int)sizeof(ks::Workspace));
  const ks::ProblemView& pv = *(const ks::ProblemView*)extra;
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  ks::Engine<ks::Wave, false> eng(pv, *(ks::Workspace*)(extra + kViewBytes), tables);
  eng.solve();
}
// A consolidation sweep over a resident cluster: block b runs the general engine on probes b, b + gridDim.x, ... — one view of
// the cluster for all of them, one workspace per probe (its claims and its node overlay), one LDS plan for the launch.
__global__ void __launch_bounds__(64) ksolve_pack_sweep(const ks::ProblemView* pv, ks::Workspace* items, int n, ks::LdsPlan plan) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, plan);
  char* const extra = lds + ((plan.total_bytes + 15) & ~15);
  ks_copy_words(extra, pv, (int)sizeof(ks::ProblemView));
  for (int p = (int)blockIdx.x; p < n; p += (int)gridDim.x) {
    ks_copy_words(extra + kViewBytes, items + p, (int)sizeof(ks::Workspace));
    ks::Engine<ks::Wave, true> eng(*(const ks::ProblemView*)extra, *(ks::Workspace*)(extra + kViewBytes), tables);
    eng.solve();
  }
}
