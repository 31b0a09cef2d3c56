// pack_kernels.h — the pack kernels (one wavefront per scheduling problem) live in translation units of their own, so that build()
// compiles them side by side (the general engine's instantiations alone are minutes of one hipcc): ksolve_pack_general.hip,
// ksolve_pack_batch.hip, ksolve_pack_sweep4.hip, ksolve_pack_fast.hip, ksolve_pack_topo.hip. ksolve.hip (the HIP backend of the C
// ABI) launches them through these declarations; no device function crosses a translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include "engine.h"
#include "fast_engine.h"
#include "topo_types.h"

__global__ void ksolve_pack(ks::ProblemView pv, ks::Workspace ws);
__global__ void ksolve_pack_lite(ks::ProblemView pv, ks::Workspace ws);
__global__ void ksolve_pack_big(ks::ProblemView pv, ks::Workspace ws);
__global__ void ksolve_pack_batch(ks::BatchItem* items);
__global__ void ksolve_pack_batch_lite(ks::BatchItem* items);
__global__ void ksolve_pack_sweep(const ks::ProblemView* pv, ks::Workspace* items, int n, ks::LdsPlan plan);
__global__ void ksolve_pack_sweep4(const ks::ProblemView* pv, ks::Workspace* items, int n, ks::LdsPlan plan, const uint32_t* order, uint32_t* next);
// the cursor engine per memory plan (g: FastPlan::global_state) and rows of class slots (r: 1 or ks::kFastRows)
__global__ void ksolve_pack_fast_g0r1(const ks::FastArgs* a);
__global__ void ksolve_pack_fast_g1r1(const ks::FastArgs* a);
__global__ void ksolve_pack_fast_g2r1(const ks::FastArgs* a);
__global__ void ksolve_pack_fast_g0r4(const ks::FastArgs* a);
__global__ void ksolve_pack_fast_g1r4(const ks::FastArgs* a);
__global__ void ksolve_pack_fast_g2r4(const ks::FastArgs* a);
__global__ void ksolve_pack_fast2(const ks::FastArgs* a);
__global__ void ksolve_pack_fast_batch(const ks::FastArgs* const* items);
__global__ void ksolve_pack_topo(const ks::TopoArgs* a);
