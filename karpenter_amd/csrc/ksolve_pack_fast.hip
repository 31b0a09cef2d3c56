// ksolve_pack_fast.hip — the cursor engine (fast_engine.h).
#include "pack_kernels.h"

// The cursor engine (fast_engine.h) for purely positive provisioning batches: one wavefront, O(1) steps. Compiled per memory plan
// (GS = 0: claim records and order in LDS; 1: records in HBM — problems that need more in-flight claims than a CU's LDS holds
// beside the caches, the host retries here when the LDS plan ran out of claims; 2: the order arrays in HBM too, up to 65,472
// in-flight claims — the exact configs[3] batch of 10M pods, 27,345) and per number of class-slot rows (FastPlan::rows).
#define KS_PACK_FAST(NAME, GS, R)                                                   \
  __global__ void __launch_bounds__(64) NAME(const ks::FastArgs* a) {               \
    extern __shared__ __attribute__((aligned(16))) char lds[];                      \
    ks::FastEngine<ks::Wave, GS, R> eng(&a->pv, &a->ws, &a->fw, lds);               \
    eng.solve();                                                                    \
  }
KS_PACK_FAST(ksolve_pack_fast_g0r1, 0, 1)
KS_PACK_FAST(ksolve_pack_fast_g1r1, 1, 1)
KS_PACK_FAST(ksolve_pack_fast_g2r1, 2, 1)
KS_PACK_FAST(ksolve_pack_fast_g0r4, 0, ks::kFastRows)
KS_PACK_FAST(ksolve_pack_fast_g1r4, 1, ks::kFastRows)
KS_PACK_FAST(ksolve_pack_fast_g2r4, 2, ks::kFastRows)
#undef KS_PACK_FAST
static_assert(ks::kFastRows == 4, "the kernels' names say four rows");
// The LDS plan with one row of class slots on TWO wavefronts (FastPlan::helper; fast_engine.h FastMail): wavefront 0 places the
// pods, another one recomputes the acceptance words of the claim a pod was added to while wavefront 0 is at the next pod. The two must
// sit on DIFFERENT SIMDs to issue side by side, and where the dispatcher puts a workgroup's wavefronts is its business: the
// workgroup comes with four, each notes its SIMD (HW_ID bits 5:4), and the first one on another SIMD than wavefront 0's stays as
// the refresher; the others leave. One barrier, in front of everything: the mailbox is zero when they part.
__global__ void __launch_bounds__(256) ksolve_pack_fast2(const ks::FastArgs* a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KS_LDS ks::FastHot* const hs = (KS_LDS ks::FastHot*)(lds + a->fw.plan.off_hot);
  const int wave = (int)(threadIdx.x >> 6);
  if (threadIdx.x == 0) ks::fast_mail_init(&hs->mail);
  if ((threadIdx.x & 63) == 0) hs->mail.simd[wave] = (uint32_t)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_REG_HW_ID, SIMD_ID
  __syncthreads();
  if (wave > 0) {
    int pick = 1;
    for (int w = 3; w >= 1; --w) if (hs->mail.simd[w] != hs->mail.simd[0]) pick = w;
    if (wave != __builtin_amdgcn_readfirstlane(pick)) return;
    ks::fast_helper_run<ks::Wave, 0, 1>(&a->fw, lds);
    return;
  }
  ks::FastEngine<ks::Wave, 0, 1, true> eng(&a->pv, &a->ws, &a->fw, lds);
  eng.solve();
  if (threadIdx.x == 0) ks::mail_store(&hs->mail.quit, 1u);
}
// Batched form: block b runs the cursor engine (LDS plan) on problem b.
__global__ void __launch_bounds__(64) ksolve_pack_fast_batch(const ks::FastArgs* const* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const ks::FastArgs* a = items[blockIdx.x];
  if (a->fw.plan.rows == 1) { ks::FastEngine<ks::Wave, 0, 1> eng(&a->pv, &a->ws, &a->fw, lds); eng.solve(); }
  else { ks::FastEngine<ks::Wave, 0, ks::kFastRows> eng(&a->pv, &a->ws, &a->fw, lds); eng.solve(); }
}
