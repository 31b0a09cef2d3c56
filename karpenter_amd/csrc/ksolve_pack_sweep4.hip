// ksolve_pack_sweep4.hip — the compact consolidation sweep: four wavefronts per workgroup on ScratchSmall.
#include "pack_kernels.h"

// The compact form of the sweep (LdsPlan::waves = 4): four wavefronts per workgroup, each on its own probes, sharing the read-only
// instance-type tables and the template records in LDS (wave 0 fills them and runs the template prefilter once, then the
// workgroup's only barrier); every wavefront keeps a ScratchSmall working set. 256 VGPRs per wavefront (two wavefronts per
// SIMD): eight probes per CU in flight instead of four — the probes are chains of dependent steps, so a launch goes as fast as
// the number of them the chip holds at once. Wave w of block b runs probes 4b + w, 4b + w + 4 * gridDim.x, ...
// (round 6) The cluster's view and each probe's workspace record are copied into LDS before an engine is built on them: the engine
// reads their fields (table pointers, sizes, option flags) all through a probe, and from HBM every such read was a vector load with
// an L2 round trip in front of its use — the records are wave-uniform, but the compiler cannot scalarise loads through a pointer it
// cannot prove read-only. ks::kSweepLdsExtra bytes behind the plan's own: the view once per workgroup, a workspace per wavefront.
// (measurement build -DKSOLVE_SWEEP4_NO_SPILL: one wavefront per SIMD, 512 VGPRs, nothing spilled — what the spills of the shipped form cost
// in traffic and time is the difference between the two, profiles/round6)
#ifdef KSOLVE_SWEEP4_NO_SPILL
#define KS_SWEEP4_BOUNDS __launch_bounds__(256, 1)
#else
#define KS_SWEEP4_BOUNDS __launch_bounds__(256, 2)
#endif
__global__ void KS_SWEEP4_BOUNDS ksolve_pack_sweep4(const ks::ProblemView* pv, ks::Workspace* items, int n, ks::LdsPlan plan, const uint32_t* order, uint32_t* next) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef ks::Engine<ks::Wave, true, false, ks::ScratchSmall> Eng;
  // (readfirstlane: the compiler takes threadIdx.x for divergent, and with it every per-wavefront LDS pointer derived from the wave's index —
  // a dozen 64-bit pointers held in vector registers across the whole probe, first in line to be spilled)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  ks::LdsTables tables;
  tables.bind(lds, plan, wave);
  uint32_t* misc = (uint32_t*)(lds + plan.off_shared_misc);
  char* const extra = lds + ((plan.total_bytes + 15) & ~15);
  ks::ProblemView* const lpv = (ks::ProblemView*)extra;
  ks::Workspace* const lws = (ks::Workspace*)(extra + ((sizeof(ks::ProblemView) + 15) & ~(size_t)15) + (size_t)wave * ((sizeof(ks::Workspace) + 15) & ~(size_t)15));
  static_assert(((sizeof(ks::ProblemView) + 15) & ~(size_t)15) + 4 * ((sizeof(ks::Workspace) + 15) & ~(size_t)15) <= (size_t)ks::kSweepLdsExtra, "kSweepLdsExtra");
  {
    const uint64_t* src = (const uint64_t*)pv;
    uint64_t* dst = (uint64_t*)lpv;
    for (int i = (int)threadIdx.x; i < (int)(sizeof(ks::ProblemView) / 8); i += 256) dst[i] = src[i];
  }
  __syncthreads();
  auto take = [&](int p) {   // probe p's workspace record into this wavefront's LDS copy
    const uint64_t* src = (const uint64_t*)(items + p);
    uint64_t* dst = (uint64_t*)lws;
    for (int i = (int)(threadIdx.x & 63); i < (int)(sizeof(ks::Workspace) / 8); i += 64) dst[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // The probes are handed out through one counter, in `order` (most displaced pods first): a wavefront that is free takes the next
  // one. With a fixed share per wavefront the launch lasted as long as its unluckiest wavefront — 2.9 ms where the mean work of
  // 2048 wavefronts over 10,000 probes is 1.9 ms. Which wavefront runs a probe does not touch its result: probes share nothing
  // but the read-only cluster.
  auto fetch = [&]() -> int {
    unsigned i = 0;
    if ((threadIdx.x & 63) == 0) i = atomicAdd(next, 1u);
    i = (unsigned)__builtin_amdgcn_readfirstlane((int)i);
    return i < (unsigned)n ? (int)order[i] : -1;
  };
  int p = -1;
  if (wave == 0) {
    p = fetch();
    if (p >= 0) {
      take(p);
      Eng eng(*lpv, *lws, tables);   // this probe's workspace lends its per-template arrays; the wavefront solves it next
      const uint32_t active = eng.prepare();
      if ((threadIdx.x & 63) == 0) misc[0] = active;
    }
  }
  __syncthreads();
  const uint32_t active = (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[0]);
  if (wave != 0) p = fetch();
  while (p >= 0) {
    take(p);
    Eng eng(*lpv, *lws, tables);
    eng.solve(&active);
    p = fetch();
  }
}
