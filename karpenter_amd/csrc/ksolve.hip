// ksolve.hip — HIP backend of the C ABI (include/ksolve.h) for MI355X (gfx950, wave64).
//
// Kernels (names as they appear in rocprofv3 traces):
//   ksolve_it_index      one thread per instance type: inverts InstanceType.Requirements into per-(key,value) bitmasks
//   ksolve_row_hash      one thread per pod row, ONE pass over the pod SoA: hash, class-table slot (read before any atomic),
//                        smallest row of the slot as representative, exact compare against an earlier row of the slot
//   ksolve_row_verify    class numbering (every slot's final representative draws an id)
//   ksolve_row_class / ksolve_class_gather   class ids per row, class tables
//   ksolve_sort_key      queue-order key extraction (5 stable LSD passes with rocprim::radix_sort_pairs)
//   ksolve_pack          the pack engine: ONE wavefront per scheduling problem (engine.h)
//   ksolve_finalize      one thread per claim: cheapest compatible available offering
// Each handle owns a stream; every phase is bracketed by HIP events recorded on that stream.
#include <cstring>
#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "ksolve_impl.h"

struct HipBackend {
  hipStream_t stream = nullptr;
  hipEvent_t ev0[8], ev1[8];
  bool failed = false;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int device = 0;
};
#define HB(h) ((HipBackend*)(h)->backend)
static bool hip_check(ksolve_handle* h, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  HB(h)->failed = true;
  if (h->error.empty()) h->error = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}

static void* be_alloc(ksolve_handle* h, size_t bytes) {
  void* p = nullptr;
  if (!hip_check(h, hipMalloc(&p, bytes ? bytes : 8), "hipMalloc")) return nullptr;
  h->allocations.push_back(p);
  hip_check(h, hipMemsetAsync(p, 0, bytes ? bytes : 8, HB(h)->stream), "hipMemsetAsync");
  return p;
}
static void be_h2d(ksolve_handle* h, void* dst, const void* src, size_t bytes) {
  if (!dst || !bytes) return;
  hip_check(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, HB(h)->stream), "hipMemcpy H2D");
  // source buffers are caller-owned and only guaranteed for the duration of the call
  hip_check(h, hipStreamSynchronize(HB(h)->stream), "hipStreamSynchronize");
}
static void be_d2h(ksolve_handle* h, void* dst, const void* src, size_t bytes) {
  if (!src || !bytes) return;
  hip_check(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, HB(h)->stream), "hipMemcpy D2H");
}
static void be_fill(ksolve_handle* h, void* dst, int byte, size_t bytes) {
  if (!dst || !bytes) return;
  hip_check(h, hipMemsetAsync(dst, byte, bytes, HB(h)->stream), "hipMemsetAsync");
}
static void be_thread_init(ksolve_handle* h) { (void)hipSetDevice(HB(h)->device); }
static void be_sync(ksolve_handle* h) { hip_check(h, hipStreamSynchronize(HB(h)->stream), "hipStreamSynchronize"); }
static bool be_ok(ksolve_handle* h) { return !HB(h)->failed; }
static void be_tic(ksolve_handle* h, int slot) { hip_check(h, hipEventRecord(HB(h)->ev0[slot], HB(h)->stream), "hipEventRecord"); }
static void be_toc(ksolve_handle* h, int slot) {
  hip_check(h, hipEventRecord(HB(h)->ev1[slot], HB(h)->stream), "hipEventRecord");
  hip_check(h, hipEventSynchronize(HB(h)->ev1[slot]), "hipEventSynchronize");
  float ms = 0;
  if (hipEventElapsedTime(&ms, HB(h)->ev0[slot], HB(h)->ev1[slot]) == hipSuccess) h->timers.ms[slot] = ms;
}

// ---- kernels ----
__global__ void ksolve_it_index(int n, ks::ItIndexArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::it_index_body(i, a);
}
__global__ void ksolve_row_hash(int n, ks::RowArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::row_hash_body(i, a);
}
__global__ void ksolve_row_verify(int n, ks::RowArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::row_verify_body(i, a);
}
__global__ void ksolve_row_class(int n, ks::RowArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::row_class_body(i, a);
}
__global__ void ksolve_class_gather(int n, ks::RowArgs a) {   // one wavefront per class
  int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i < n) ks::class_gather_body(i, a, (int)(threadIdx.x & 63), 64);
}
__global__ void ksolve_sort_key(int n, ks::SortKeyArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::sort_key_body(i, a);
}
__global__ void ksolve_iota(int n, uint32_t* idx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = (uint32_t)i;
}
__global__ void ksolve_finalize(int n, ks::FinalizeArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::finalize_body(i, a);
}
// One wavefront (64 threads) per problem; the working requirement set and candidate lists live in LDS.
__global__ void __launch_bounds__(64) ksolve_pack(ks::ProblemView pv, ks::Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  ks::Engine<ks::Wave, true> eng(pv, ws, tables);
  eng.solve();
}
// The same engine compiled without topology / existing nodes / daemon overhead / minValues / reservations, for problems
// that use none of them (ProblemView::lite): less code and far less live state around the hot loop.
__global__ void __launch_bounds__(64) ksolve_pack_lite(ks::ProblemView pv, ks::Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  ks::Engine<ks::Wave, false> eng(pv, ws, tables);
  eng.solve();
}

// The full engine with the claim order in HBM, for problems with more in-flight claims than a CU's LDS can order
// (every anti-affinity / hostname-spread pod is its own NodeClaim): ProblemView::big.
__global__ void __launch_bounds__(64) ksolve_pack_big(ks::ProblemView pv, ks::Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  ks::Engine<ks::Wave, true, true> eng(pv, ws, tables);
  eng.solve();
}
// Batched form: block b solves problem b (its view and workspace are read from HBM instead of the kernel arguments).
__global__ void __launch_bounds__(64) ksolve_pack_batch(ks::BatchItem* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::BatchItem& it = items[blockIdx.x];
  ks::LdsTables tables;
  tables.bind(lds, it.pv.lds);
  ks::Engine<ks::Wave, true> eng(it.pv, it.ws, tables);
  eng.solve();
}
__global__ void __launch_bounds__(64) ksolve_pack_batch_lite(ks::BatchItem* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::BatchItem& it = items[blockIdx.x];
  ks::LdsTables tables;
  tables.bind(lds, it.pv.lds);
  ks::Engine<ks::Wave, false> eng(it.pv, it.ws, tables);
  eng.solve();
}
// The cursor engine (fast_engine.h) for purely positive provisioning batches: one wavefront, O(1) steps.
__global__ void __launch_bounds__(64) ksolve_pack_fast(const ks::FastArgs* a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::FastEngine<ks::Wave> eng(&a->pv, &a->ws, &a->fw, lds);
  eng.solve();
}
// Batched form: block b runs the cursor engine on problem b.
__global__ void __launch_bounds__(64) ksolve_pack_fast_batch(const ks::FastArgs* const* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const ks::FastArgs* a = items[blockIdx.x];
  ks::FastEngine<ks::Wave> eng(&a->pv, &a->ws, &a->fw, lds);
  eng.solve();
}
__global__ void ksolve_fast_queue(int n, ks::FastQueueArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::fast_queue_body(i, a);
}
__global__ void ksolve_fast_scatter(int n, ks::FastQueueArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::fast_scatter_body(i, a);
}
// One wavefront per claim: hot claim records (requirement masks, InstanceTypeOptions) from the cursor engine's compact state.
__global__ void __launch_bounds__(64) ksolve_fast_records(ks::FastRecordArgs a) {
  ks::fast_record_body<ks::Wave>((int)blockIdx.x, a);
}
static dim3 grid_for(int n) { return dim3((unsigned)((n + 255) / 256)); }
static void be_launch_it_index(ksolve_handle* h, int n, const ks::ItIndexArgs& a) { hipLaunchKernelGGL(ksolve_it_index, grid_for(n), dim3(256), 0, HB(h)->stream, n, a); }
static void be_launch_row_hash(ksolve_handle* h, int n, const ks::RowArgs& a) { hipLaunchKernelGGL(ksolve_row_hash, grid_for(n), dim3(256), 0, HB(h)->stream, n, a); }
static void be_launch_row_verify(ksolve_handle* h, int n, const ks::RowArgs& a) { hipLaunchKernelGGL(ksolve_row_verify, grid_for(n), dim3(256), 0, HB(h)->stream, n, a); }
static void be_launch_row_class(ksolve_handle* h, int n, const ks::RowArgs& a) { hipLaunchKernelGGL(ksolve_row_class, grid_for(n), dim3(256), 0, HB(h)->stream, n, a); }
static void be_launch_class_gather(ksolve_handle* h, int n, const ks::RowArgs& a) { hipLaunchKernelGGL(ksolve_class_gather, grid_for(n * 64), dim3(256), 0, HB(h)->stream, n, a); }
static void be_launch_finalize(ksolve_handle* h, int n, const ks::FinalizeArgs& a) { hipLaunchKernelGGL(ksolve_finalize, grid_for(n), dim3(256), 0, HB(h)->stream, n, a); }
static void be_launch_pack(ksolve_handle* h) {
  const int lds_bytes = h->pv.lds.total_bytes;
  const void* fn = h->pv.big ? (const void*)ksolve_pack_big : h->pv.lite ? (const void*)ksolve_pack_lite : (const void*)ksolve_pack;
  if (!hip_check(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)")) return;
  if (h->pv.big) hipLaunchKernelGGL(ksolve_pack_big, dim3(1), dim3(64), (size_t)lds_bytes, HB(h)->stream, h->pv, h->ws);
  else if (h->pv.lite) hipLaunchKernelGGL(ksolve_pack_lite, dim3(1), dim3(64), (size_t)lds_bytes, HB(h)->stream, h->pv, h->ws);
  else hipLaunchKernelGGL(ksolve_pack, dim3(1), dim3(64), (size_t)lds_bytes, HB(h)->stream, h->pv, h->ws);
  hip_check(h, hipGetLastError(), "ksolve_pack launch");
}

static void be_launch_pack_fast(ksolve_handle* h) {
  const int lds_bytes = h->fw.plan.total_bytes;
  if (!hip_check(h, hipFuncSetAttribute((const void*)ksolve_pack_fast, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)")) return;
  ks::FastArgs a{h->pv, h->ws, h->fw};
  be_h2d(h, h->d_fast_args, &a, sizeof(a));
  hipLaunchKernelGGL(ksolve_pack_fast, dim3(1), dim3(64), (size_t)lds_bytes, HB(h)->stream, (const ks::FastArgs*)h->d_fast_args);
  hip_check(h, hipGetLastError(), "ksolve_pack_fast launch");
}
static void be_launch_pack_fast_batch(ksolve_handle** hs, int n) {
  if (n <= 0) return;
  ksolve_handle* h0 = hs[0];
  HipBackend* b = HB(h0);
  std::vector<const ks::FastArgs*> ptrs((size_t)n);
  int lds_bytes = 0;
  for (int i = 0; i < n; ++i) {
    ks::FastArgs a{hs[i]->pv, hs[i]->ws, hs[i]->fw};
    be_h2d(hs[i], hs[i]->d_fast_args, &a, sizeof(a));
    ptrs[(size_t)i] = hs[i]->d_fast_args;
    lds_bytes = std::max(lds_bytes, hs[i]->fw.plan.total_bytes);
  }
  const ks::FastArgs** d_ptrs = nullptr;
  if (!hip_check(h0, hipMalloc((void**)&d_ptrs, (size_t)n * sizeof(void*)), "hipMalloc(batch)")) return;
  hip_check(h0, hipMemcpyAsync(d_ptrs, ptrs.data(), (size_t)n * sizeof(void*), hipMemcpyHostToDevice, b->stream), "hipMemcpy(batch)");
  hip_check(h0, hipStreamSynchronize(b->stream), "hipStreamSynchronize");
  if (hip_check(h0, hipFuncSetAttribute((const void*)ksolve_pack_fast_batch, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)")) {
    hip_check(h0, hipEventRecord(b->ev0[ksi::T_PACK], b->stream), "hipEventRecord");
    hipLaunchKernelGGL(ksolve_pack_fast_batch, dim3((unsigned)n), dim3(64), (size_t)lds_bytes, b->stream, (const ks::FastArgs* const*)d_ptrs);
    hip_check(h0, hipGetLastError(), "ksolve_pack_fast_batch launch");
    hip_check(h0, hipEventRecord(b->ev1[ksi::T_PACK], b->stream), "hipEventRecord");
    hip_check(h0, hipEventSynchronize(b->ev1[ksi::T_PACK]), "hipEventSynchronize");
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->ev0[ksi::T_PACK], b->ev1[ksi::T_PACK]) == hipSuccess) for (int i = 0; i < n; ++i) hs[i]->timers.ms[ksi::T_PACK] = ms;
    if (b->failed) for (int i = 1; i < n; ++i) { HB(hs[i])->failed = true; hs[i]->error = h0->error; }
  }
  (void)hipFree(d_ptrs);
}
static ks::FastQueueArgs fast_queue_args(ksolve_handle* h) {
  return ks::FastQueueArgs{h->pv.sorted_pods, h->pv.row_class, h->fw.q_class, h->fw.q_claim, h->fw.q_cnt, h->ws.assign, h->ws.slot};
}
static void be_launch_fast_records(ksolve_handle* h, int n_claims) {
  ks::FastRecordArgs a{h->pv, h->ws, h->fw};
  hipLaunchKernelGGL(ksolve_fast_records, dim3((unsigned)n_claims), dim3(64), 0, HB(h)->stream, a);
  hip_check(h, hipGetLastError(), "ksolve_fast_records launch");
  const int n = (int)h->n_pods;
  hipLaunchKernelGGL(ksolve_fast_scatter, grid_for(n), dim3(256), 0, HB(h)->stream, n, fast_queue_args(h));
  hip_check(h, hipGetLastError(), "ksolve_fast_scatter launch");
}
static void be_launch_fast_queue(ksolve_handle* h) {
  const int n = (int)h->n_pods;
  hipLaunchKernelGGL(ksolve_fast_queue, grid_for(n), dim3(256), 0, HB(h)->stream, n, fast_queue_args(h));
  hip_check(h, hipGetLastError(), "ksolve_fast_queue launch");
}

// One launch per engine flavour (lite / full problems of the batch), each on the stream of its first handle so that the
// two launches overlap; every handle's T_PACK timer gets the time of its own launch.
static void launch_batch_group(std::vector<ksolve_handle*>& g, bool lite, ks::BatchItem** d_items_out) {
  *d_items_out = nullptr;
  if (g.empty()) return;
  ksolve_handle* h0 = g[0];
  HipBackend* b = HB(h0);
  std::vector<ks::BatchItem> items(g.size());
  int lds_bytes = 0;
  for (size_t i = 0; i < g.size(); ++i) { items[i].pv = g[i]->pv; items[i].ws = g[i]->ws; lds_bytes = std::max(lds_bytes, g[i]->pv.lds.total_bytes); }
  ks::BatchItem* d_items = nullptr;
  if (!hip_check(h0, hipMalloc((void**)&d_items, items.size() * sizeof(ks::BatchItem)), "hipMalloc(batch)")) return;
  *d_items_out = d_items;
  hip_check(h0, hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(ks::BatchItem), hipMemcpyHostToDevice, b->stream), "hipMemcpy(batch)");
  hip_check(h0, hipStreamSynchronize(b->stream), "hipStreamSynchronize");   // `items` is a local
  const void* fn = lite ? (const void*)ksolve_pack_batch_lite : (const void*)ksolve_pack_batch;
  if (!hip_check(h0, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)")) return;
  hip_check(h0, hipEventRecord(b->ev0[ksi::T_PACK], b->stream), "hipEventRecord");
  if (lite) hipLaunchKernelGGL(ksolve_pack_batch_lite, dim3((unsigned)g.size()), dim3(64), (size_t)lds_bytes, b->stream, d_items);
  else hipLaunchKernelGGL(ksolve_pack_batch, dim3((unsigned)g.size()), dim3(64), (size_t)lds_bytes, b->stream, d_items);
  hip_check(h0, hipGetLastError(), "ksolve_pack_batch launch");
  hip_check(h0, hipEventRecord(b->ev1[ksi::T_PACK], b->stream), "hipEventRecord");
}
static void finish_batch_group(std::vector<ksolve_handle*>& g, ks::BatchItem* d_items) {
  if (g.empty()) return;
  ksolve_handle* h0 = g[0];
  HipBackend* b = HB(h0);
  hip_check(h0, hipEventSynchronize(b->ev1[ksi::T_PACK]), "hipEventSynchronize");
  float ms = 0;
  if (hipEventElapsedTime(&ms, b->ev0[ksi::T_PACK], b->ev1[ksi::T_PACK]) == hipSuccess) h0->timers.ms[ksi::T_PACK] = ms;
  for (size_t i = 1; i < g.size(); ++i) { g[i]->timers.ms[ksi::T_PACK] = h0->timers.ms[ksi::T_PACK]; if (b->failed) { HB(g[i])->failed = true; g[i]->error = h0->error; } }
  if (d_items) (void)hipFree(d_items);
}
static void be_launch_pack_batch(ksolve_handle** hs, int n) {
  std::vector<ksolve_handle*> lite, full;
  for (int i = 0; i < n; ++i) {
    if (hs[i]->pv.big) { be_tic(hs[i], ksi::T_PACK); be_launch_pack(hs[i]); be_toc(hs[i], ksi::T_PACK); continue; }   // BIG problems run alone
    (hs[i]->pv.lite ? lite : full).push_back(hs[i]);
  }
  ks::BatchItem *dl = nullptr, *df = nullptr;
  launch_batch_group(lite, true, &dl);
  launch_batch_group(full, false, &df);
  finish_batch_group(lite, dl);
  finish_batch_group(full, df);
}
// queue order (queue.go:72-108): five stable LSD radix passes over 64-bit keys, least significant criterion first
static void be_sort_pods(ksolve_handle* h) {
  const int n = (int)h->n_pods;
  HipBackend* b = HB(h);
  uint32_t *in = h->d_idx_a, *out = h->d_idx_b;
  h->pv.sorted_pods = in;
  if (n == 0) return;
  hipLaunchKernelGGL(ksolve_iota, grid_for(n), dim3(256), 0, b->stream, n, in);
  for (int pass = 0; pass < 5; ++pass) {
    ks::SortKeyArgs a = h->sort_args;
    a.idx_in = in; a.key_out = h->d_key_a; a.pass = pass;
    hipLaunchKernelGGL(ksolve_sort_key, grid_for(n), dim3(256), 0, b->stream, n, a);
    size_t need = 0;
    hip_check(h, rocprim::radix_sort_pairs(nullptr, need, h->d_key_a, h->d_key_b, in, out, (size_t)n, 0, 64, b->stream), "radix_sort_pairs(size)");
    if (need > b->sort_tmp_bytes) {
      if (b->sort_tmp) (void)hipFree(b->sort_tmp);
      b->sort_tmp = nullptr;
      if (!hip_check(h, hipMalloc(&b->sort_tmp, need), "hipMalloc(sort)")) return;
      b->sort_tmp_bytes = need;
    }
    hip_check(h, rocprim::radix_sort_pairs(b->sort_tmp, need, h->d_key_a, h->d_key_b, in, out, (size_t)n, 0, 64, b->stream), "radix_sort_pairs");
    std::swap(in, out);
  }
  h->pv.sorted_pods = in;
}

static int be_device_available() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
  return std::string(p.gcnArchName).rfind("gfx950", 0) == 0 ? 1 : 0;
}

extern "C" {

ksolve_status ksolve_create(const ksolve_problem_desc* desc, const ksolve_options* opts, ksolve_handle** out) {
  ksolve_handle* h = new ksolve_handle();
  HipBackend* b = new HipBackend();
  h->backend = b;
  *out = h;
  if (!be_device_available()) { h->error = "no usable gfx950 device (hipGetDeviceCount/hipGetDeviceProperties)"; return KSOLVE_ERR_NO_DEVICE; }
  b->device = opts ? (int)opts->device : 0;
  if (!hip_check(h, hipSetDevice(b->device), "hipSetDevice")) return KSOLVE_ERR_DEVICE;
  if (!hip_check(h, hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking), "hipStreamCreate")) return KSOLVE_ERR_DEVICE;
  for (int i = 0; i < 8; ++i) { hip_check(h, hipEventCreate(&b->ev0[i]), "hipEventCreate"); hip_check(h, hipEventCreate(&b->ev1[i]), "hipEventCreate"); }
  return ksi::create(desc, opts, h);
}
ksolve_status ksolve_probe_create(ksolve_handle* base, const ksolve_probe* probe, ksolve_handle** out) {
  if (!out) return KSOLVE_ERR_INVALID;
  ksolve_handle* h = new ksolve_handle();
  HipBackend* b = new HipBackend();
  h->backend = b;
  *out = h;
  if (!base || !base->backend) { h->error = "null base handle"; return KSOLVE_ERR_INVALID; }
  b->device = HB(base)->device;
  if (!hip_check(h, hipSetDevice(b->device), "hipSetDevice")) return KSOLVE_ERR_DEVICE;
  if (!hip_check(h, hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking), "hipStreamCreate")) return KSOLVE_ERR_DEVICE;
  for (int i = 0; i < 8; ++i) { hip_check(h, hipEventCreate(&b->ev0[i]), "hipEventCreate"); hip_check(h, hipEventCreate(&b->ev1[i]), "hipEventCreate"); }
  return ksi::probe_create(base, probe, h);
}
ksolve_status ksolve_solve(ksolve_handle* h, ksolve_results* out) {
  if (!h || !h->backend) return KSOLVE_ERR_INVALID;
  if (hipSetDevice(HB(h)->device) != hipSuccess) return KSOLVE_ERR_DEVICE;
  return ksi::solve(h, out);
}
ksolve_status ksolve_solve_batch(ksolve_handle** hs, uint32_t n, ksolve_results* outs) {
  if (!hs || !outs || n == 0) return KSOLVE_ERR_INVALID;
  for (uint32_t i = 0; i < n; ++i) if (!hs[i] || !hs[i]->backend || HB(hs[i])->device != HB(hs[0])->device) return KSOLVE_ERR_INVALID;
  if (hipSetDevice(HB(hs[0])->device) != hipSuccess) return KSOLVE_ERR_DEVICE;
  return ksi::solve_batch(hs, n, outs);
}
ksolve_status ksolve_cancel(ksolve_handle* h) {
  if (!h || !h->d_cancel) return KSOLVE_ERR_INVALID;
  int one = 1;
  // written from another thread while the pack kernel polls the flag between pods
  (void)hipMemcpy(h->d_cancel, &one, sizeof(int), hipMemcpyHostToDevice);
  return KSOLVE_OK;
}
void ksolve_results_free(ksolve_results* r) {
  if (r && r->impl) { delete (ksi::ResultsImpl*)r->impl; r->impl = nullptr; }
}
void ksolve_destroy(ksolve_handle* h) {
  if (!h) return;
  HipBackend* b = HB(h);
  if (b) {
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void* p : h->allocations) (void)hipFree(p);
    if (b->sort_tmp) (void)hipFree(b->sort_tmp);
    if (b->stream) { for (int i = 0; i < 8; ++i) { (void)hipEventDestroy(b->ev0[i]); (void)hipEventDestroy(b->ev1[i]); } (void)hipStreamDestroy(b->stream); }
    delete b;
  }
  delete h;
}
const char* ksolve_last_error(const ksolve_handle* h) { return h ? h->error.c_str() : "null handle"; }
uint32_t ksolve_abi_version(void) { return KSOLVE_ABI_VERSION; }
int ksolve_device_available(void) { return be_device_available(); }
double ksolve_last_kernel_ms(const ksolve_handle* h, const char* name) {
  if (!h) return -1;
  std::string n(name ? name : "");
  if (n == "ksolve_pack") return h->timers.ms[ksi::T_PACK];
  if (n == "classify") return h->timers.ms[ksi::T_CLASSIFY];
  if (n == "sort") return h->timers.ms[ksi::T_SORT];
  if (n == "it_index") return h->timers.ms[ksi::T_INDEX];
  return -1;
}

}  // extern "C"
