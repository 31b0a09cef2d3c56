// ksolve.hip — HIP backend of the C ABI (include/ksolve.h) for MI355X (gfx950, wave64).
//
// Kernels (names as they appear in rocprofv3 traces):
//   ksolve_it_index      one thread per instance type: inverts InstanceType.Requirements into per-(key,value) bitmasks
//   ksolve_row_hash      one thread per pod row, ONE pass over the pod SoA: hash, class-table slot (read before any atomic),
//                        smallest row of the slot as representative, exact compare against an earlier row of the slot
//   ksolve_row_class / ksolve_class_gather   class ids per row, class tables
//   ksolve_sort_key      queue-order key extraction (5 stable LSD passes with rocprim::radix_sort_pairs)
//   ksolve_pack          the pack engine: ONE wavefront per scheduling problem (engine.h)
//   ksolve_finalize      one thread per claim: cheapest compatible available offering
// Each handle owns a stream; every phase is bracketed by HIP events recorded on that stream.
#include <cstring>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <rocprim/rocprim.hpp>

#include "ksolve_impl.h"
#include "pack_kernels.h"

struct HipBackend {
  hipStream_t stream = nullptr;
  hipEvent_t ev0[8], ev1[8];
  bool failed = false;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  void* stage = nullptr;        // pinned host staging of a sweep's descriptors (be_stage)
  size_t stage_bytes = 0;
  int device = 0;
  int n_cus = 256;              // compute units of the device (the compact sweep launches what the chip holds at once)
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
static void* be_try_alloc(ksolve_handle* h, size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { (void)hipGetLastError(); return nullptr; }   // refused: the error is taken off the runtime's slate, the handle stays usable
  h->allocations.push_back(p);
  hip_check(h, hipMemsetAsync(p, 0, bytes ? bytes : 8, HB(h)->stream), "hipMemsetAsync");
  return p;
}
// Page-locked host memory the handle keeps between calls (grown on demand): what a sweep writes its per-probe workspace records and
// probe lists into, so that their upload is one DMA at link speed instead of a staged copy out of pageable memory (10,000 probes:
// 6 MB of records, 1.5 of the call's 1.9 ms of upload).
static void* be_stage(ksolve_handle* h, size_t bytes) {
  HipBackend* b = HB(h);
  if (bytes > b->stage_bytes) {
    if (b->stage) (void)hipHostFree(b->stage);
    b->stage = nullptr; b->stage_bytes = 0;
    const size_t want = bytes + bytes / 4 + (1u << 20);
    if (!hip_check(h, hipHostMalloc(&b->stage, want, hipHostMallocDefault), "hipHostMalloc(stage)")) return nullptr;
    b->stage_bytes = want;
  }
  return b->stage;
}
// Large uploads out of the caller's pageable tables (a million pod rows: 220-245 MB) go through two page-locked buffers the PROCESS
// keeps: the host copies chunk i + 1 into one while the DMA engine reads chunk i out of the other. The runtime's own pageable path
// moved the 245 MB of the headline problem at 2.4 GB/s (100 ms of a 210 ms NewScheduler); page-locking the caller's memory for the
// copy (hipHostRegister, the round-4 attempt) costs as much as it saves. Portable memory: handles on any device of the process use it,
// one at a time.
struct H2dPipe {
  std::mutex mu;
  void* buf[2] = {nullptr, nullptr};
  static constexpr size_t kChunk = (size_t)16 << 20;
  bool tried = false, ok = false;
};
static H2dPipe& h2d_pipe() { static H2dPipe p; return p; }
static bool h2d_staged(ksolve_handle* h, void* dst, const void* src, size_t bytes) {
  H2dPipe& P = h2d_pipe();
  std::lock_guard<std::mutex> guard(P.mu);
  if (!P.tried) {
    P.tried = true;
    P.ok = hipHostMalloc(&P.buf[0], H2dPipe::kChunk, hipHostMallocPortable) == hipSuccess && hipHostMalloc(&P.buf[1], H2dPipe::kChunk, hipHostMallocPortable) == hipSuccess;
    if (!P.ok) { (void)hipGetLastError(); for (auto& b : P.buf) { if (b) (void)hipHostFree(b); b = nullptr; } }
  }
  if (!P.ok) return false;
  hipStream_t st = HB(h)->stream;
  hipEvent_t ev[2];
  if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(ev[0]); return false; }
  bool used[2] = {false, false}, good = true;
  size_t off = 0;
  for (int i = 0; off < bytes && good; i ^= 1) {
    const size_t n = std::min(H2dPipe::kChunk, bytes - off);
    if (used[i]) good = hip_check(h, hipEventSynchronize(ev[i]), "hipEventSynchronize (upload staging)");   // the DMA that read this buffer is done
    if (!good) break;
    {
      // the host side of the pipeline on four threads: one thread copies cold pageable memory at 4-6 GB/s, the link takes 25+
      const size_t part = ((n / 4) + 4095) & ~(size_t)4095;
      char* d = (char*)P.buf[i]; const char* sp = (const char*)src + off;
      std::thread helpers[3];
      int nh = 0;
      for (int t = 1; t < 4; ++t) { const size_t a = (size_t)t * part; if (a >= n) break; const size_t len = std::min(part, n - a); helpers[nh++] = std::thread([=] { memcpy(d + a, sp + a, len); }); }
      memcpy(d, sp, std::min(part, n));
      for (int t = 0; t < nh; ++t) helpers[t].join();
    }
    good = hip_check(h, hipMemcpyAsync((char*)dst + off, P.buf[i], n, hipMemcpyHostToDevice, st), "hipMemcpy H2D (staged)") &&
           hip_check(h, hipEventRecord(ev[i], st), "hipEventRecord (upload staging)");
    used[i] = true;
    off += n;
  }
  hip_check(h, hipStreamSynchronize(st), "hipStreamSynchronize");   // the buffers are free again, the caller's source is no longer read
  (void)hipEventDestroy(ev[0]); (void)hipEventDestroy(ev[1]);
  return true;   // errors are on the handle (hip_check)
}
static void be_h2d(ksolve_handle* h, void* dst, const void* src, size_t bytes) {
  if (!dst || !bytes) return;
#ifdef KSOLVE_TEST_HOOKS
  static const bool no_stage = getenv("KSOLVE_TEST_NO_HOST_REGISTER") != nullptr;   // tests / A-B runs: the runtime's own pageable path
#else
  const bool no_stage = false;
#endif
  if (bytes >= (size_t)8 << 20 && !no_stage) {
    // page-locked already (the handle's own staging memory of a large sweep): one DMA at link speed as it is
    hipPointerAttribute_t at{};
    const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) { (void)hipGetLastError(); if (h2d_staged(h, dst, src, bytes)) return; }
  }
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
// Phase ranges for `rocprofv3 --marker-trace` (ROCTx; the reference's analogue is the pprof endpoint around its controllers,
// pkg/operator/operator.go:209-224): every timed phase of a solve / sweep is a named range on the calling thread. The library is
// looked up at run time — a box without libroctx64 runs without ranges.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    // rocprofv3 --marker-trace listens to the rocprofiler-sdk build of ROCTx; roctracer's libroctx64 is the fallback for older tools
    void* l = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!l) l = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
    if (!l) l = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!l) l = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_GLOBAL);
    if (l) { push = (int (*)(const char*))dlsym(l, "roctxRangePushA"); pop = (int (*)())dlsym(l, "roctxRangePop"); }
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
static RoctxApi& roctx() { static RoctxApi r; return r; }
static const char* const kPhaseRange[8] = {"ksolve:upload", "ksolve:instance_type_index / node_dead0", "ksolve:classing", "ksolve:queue_sort", "ksolve:pack", "ksolve:finalize", "ksolve:download", "ksolve:row_hash"};
static void be_tic(ksolve_handle* h, int slot) {
  if (roctx().push) roctx().push(kPhaseRange[slot & 7]);
  hip_check(h, hipEventRecord(HB(h)->ev0[slot], HB(h)->stream), "hipEventRecord");
}
static void be_range_drop(ksolve_handle*) { if (roctx().pop) roctx().pop(); }
static void be_toc(ksolve_handle* h, int slot) {
  hip_check(h, hipEventRecord(HB(h)->ev1[slot], HB(h)->stream), "hipEventRecord");
  hip_check(h, hipEventSynchronize(HB(h)->ev1[slot]), "hipEventSynchronize");
  float ms = 0;
  if (hipEventElapsedTime(&ms, HB(h)->ev0[slot], HB(h)->ev1[slot]) == hipSuccess) h->timers.ms[slot] = ms;
  if (roctx().pop) roctx().pop();
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
// The same pass with the rows' mask words staged through LDS: one wavefront per 64 consecutive rows; the 64 x req_words
// words of each table are contiguous in HBM, so lane l loads words l, l+64, ... (512 bytes per access, fully coalesced) and
// drops them at [row][word] in LDS (odd row stride: the per-row reads that follow spread over the banks); every lane then
// hashes its own row exactly as row_hash_body does.
__global__ void __launch_bounds__(64) ksolve_row_hash_coop(int n, ks::RowArgs a, int rw, uint32_t magic, uint32_t kmagic) {
  extern __shared__ __attribute__((aligned(16))) uint64_t coop_lds[];
  const int l = (int)threadIdx.x;
  const int row0 = (int)blockIdx.x * 64;
  const int rows = n - row0 < 64 ? n - row0 : 64;
  const int stride = rw | 1;
  const int total = rows * rw;
  uint64_t* t0 = coop_lds;
  uint64_t* t1 = coop_lds + 64 * stride;
  const uint64_t* g0 = a.reqs.mask + (size_t)row0 * rw;
  const uint64_t* g1 = a.strict.mask + (size_t)row0 * rw;
  for (int e = l; e < total; e += 256) {
    // four accesses of each table in flight per lane and round
    const int e1 = e + 64, e2 = e + 128, e3 = e + 192;
    const uint64_t a0 = g0[e], b0 = g1[e];
    const uint64_t a1 = e1 < total ? g0[e1] : 0, b1 = e1 < total ? g1[e1] : 0;
    const uint64_t a2 = e2 < total ? g0[e2] : 0, b2 = e2 < total ? g1[e2] : 0;
    const uint64_t a3 = e3 < total ? g0[e3] : 0, b3 = e3 < total ? g1[e3] : 0;
    { const int r = (int)__umulhi((uint32_t)e, magic); const int x = r * stride + (e - r * rw); t0[x] = a0; t1[x] = b0; }
    if (e1 < total) { const int r = (int)__umulhi((uint32_t)e1, magic); const int x = r * stride + (e1 - r * rw); t0[x] = a1; t1[x] = b1; }
    if (e2 < total) { const int r = (int)__umulhi((uint32_t)e2, magic); const int x = r * stride + (e2 - r * rw); t0[x] = a2; t1[x] = b2; }
    if (e3 < total) { const int r = (int)__umulhi((uint32_t)e3, magic); const int x = r * stride + (e3 - r * rw); t0[x] = a3; t1[x] = b3; }
  }
  // minValues tables ([rows][n_keys] int32, read once per defined key by the hash): staged the same way
  const int nk = a.dict.n_keys, kstride = nk | 1;
  int32_t* m0 = (int32_t*)(coop_lds + 2 * 64 * stride);
  int32_t* m1 = m0 + 64 * kstride;
  const int ktotal = rows * nk;
  if (a.reqs.minv) { const int32_t* g = a.reqs.minv + (size_t)row0 * nk; for (int e = l; e < ktotal; e += 64) { const int r = (int)__umulhi((uint32_t)e, kmagic); m0[r * kstride + (e - r * nk)] = g[e]; } }
  if (a.strict.minv) { const int32_t* g = a.strict.minv + (size_t)row0 * nk; for (int e = l; e < ktotal; e += 64) { const int r = (int)__umulhi((uint32_t)e, kmagic); m1[r * kstride + (e - r * nk)] = g[e]; } }
  // requests and the toleration mask of the 64 rows: [row][n_res + 1]
  const int nr = a.n_res, rstride = nr + 1;
  uint64_t* rqs = (uint64_t*)(m1 + 64 * kstride + ((64 * kstride) & 1));
  const int row = row0 + l;
  const bool live = row < n;
  const int rowc = live ? row : n - 1;
  for (int r = 0; r < nr; ++r) rqs[l * rstride + r] = (uint64_t)a.requests[(size_t)r * a.n_rows + rowc];
  rqs[l * rstride + nr] = a.tolerates[rowc];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  ks::ReqRef q = a.reqs.at(a.dict, (uint32_t)rowc), qs = a.strict.at(a.dict, (uint32_t)rowc);
  q.mask = t0 + (size_t)l * stride;
  qs.mask = t1 + (size_t)l * stride;
  if (q.minv) q.minv = m0 + (size_t)l * kstride;
  if (qs.minv) qs.minv = m1 + (size_t)l * kstride;
  const uint64_t h = live ? ks::row_hash_kept(a, ks::row_hash_value(row, a, q, qs)) : 0ull;
  // The class table is one address per class for a million rows: every device-wide access to it is made ONCE per distinct
  // hash of the wavefront (neighbouring pods mostly share their class), by the first lane that carries it. The other lanes
  // take its slot and have to equal ITS row — both rows are in LDS. It has to equal the row that reached the slot before it
  // (equality is transitive): that row's words are fetched by the whole wavefront, one lane per word (coalesced), and
  // compared with the LDS copy — no lane walks a row of HBM word by word.
  uint64_t todo = __ballot(live ? 1 : 0);
  bool bad = false;
  while (todo) {
    const int j = __builtin_ctzll(todo);
    const uint32_t hlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, j), hhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), j);
    const uint64_t hj = (uint64_t)hlo | ((uint64_t)hhi << 32);
    const bool mine = live && h == hj;
    todo &= ~__ballot(mine ? 1 : 0);
    uint32_t slot = 0, other = 0xFFFFFFFFu;
    if (l == j) other = ks::row_table_insert(row, a, h, &slot);
    slot = (uint32_t)__builtin_amdgcn_readlane((int)slot, j);
    const uint32_t rep = (uint32_t)__builtin_amdgcn_readlane((int)other, j);
    // the first lane's requirement sets, wave-uniform flags and LDS words
    ks::ReqRef qj, qsj;
    qj.defined = (uint32_t)__builtin_amdgcn_readlane((int)q.defined, j); qj.complement = (uint32_t)__builtin_amdgcn_readlane((int)q.complement, j);
    qj.has_gte = (uint32_t)__builtin_amdgcn_readlane((int)q.has_gte, j); qj.has_lte = (uint32_t)__builtin_amdgcn_readlane((int)q.has_lte, j);
    qsj.defined = (uint32_t)__builtin_amdgcn_readlane((int)qs.defined, j); qsj.complement = (uint32_t)__builtin_amdgcn_readlane((int)qs.complement, j);
    qsj.has_gte = (uint32_t)__builtin_amdgcn_readlane((int)qs.has_gte, j); qsj.has_lte = (uint32_t)__builtin_amdgcn_readlane((int)qs.has_lte, j);
    const size_t rowj = (size_t)(row0 + j);
    qj.mask = t0 + (size_t)j * stride; qsj.mask = t1 + (size_t)j * stride;
    qj.minv = a.reqs.minv ? m0 + (size_t)j * kstride : nullptr; qsj.minv = a.strict.minv ? m1 + (size_t)j * kstride : nullptr;
    qj.gte = a.reqs.gte ? a.reqs.gte + rowj * nk : nullptr; qj.lte = a.reqs.lte ? a.reqs.lte + rowj * nk : nullptr;
    qsj.gte = a.strict.gte ? a.strict.gte + rowj * nk : nullptr; qsj.lte = a.strict.lte ? a.strict.lte + rowj * nk : nullptr;
    if (mine) {
      a.row_slot[row] = slot;
      if (l != j) {
        uint64_t d = ks::reqset_diff(a.dict, q, qj) | ks::reqset_diff(a.dict, qs, qsj);
        for (int r = 0; r <= nr; ++r) d |= rqs[l * rstride + r] ^ rqs[j * rstride + r];
        if (a.host_ports) d |= (a.host_ports[(size_t)row * 2] ^ a.host_ports[rowj * 2]) | (a.host_ports[(size_t)row * 2 + 1] ^ a.host_ports[rowj * 2 + 1]);
        if (a.vol) d |= a.vol[row] ^ a.vol[rowj];
        if (a.topo_owned) for (int w = 0; w < a.topo_words; ++w) {
          d |= a.topo_owned[(size_t)row * a.topo_words + w] ^ a.topo_owned[rowj * a.topo_words + w];
          d |= a.topo_selected[(size_t)row * a.topo_words + w] ^ a.topo_selected[rowj * a.topo_words + w];
        }
        if (d) bad = true;
      }
    }
    if (rep != 0xFFFFFFFFu) {
      uint64_t d = 0;
      d |= (uint64_t)((a.reqs.defined[rep] ^ qj.defined) | (a.reqs.complement[rep] ^ qj.complement) | ((a.reqs.has_gte ? a.reqs.has_gte[rep] : 0u) ^ qj.has_gte) | ((a.reqs.has_lte ? a.reqs.has_lte[rep] : 0u) ^ qj.has_lte));
      d |= (uint64_t)((a.strict.defined[rep] ^ qsj.defined) | (a.strict.complement[rep] ^ qsj.complement) | ((a.strict.has_gte ? a.strict.has_gte[rep] : 0u) ^ qsj.has_gte) | ((a.strict.has_lte ? a.strict.has_lte[rep] : 0u) ^ qsj.has_lte));
      if (l <= nr) d |= (l < nr ? (uint64_t)a.requests[(size_t)l * a.n_rows + rep] : a.tolerates[rep]) ^ rqs[j * rstride + l];
      for (int w = l; w < rw; w += 64) {   // one lane per mask word; words of keys the set does not define are ignored (equal_reqset)
        int kw = 0;
        for (int k = 0; k < nk; ++k) kw += (int)((uint32_t)w >= a.dict.key_word_off[k + 1]);
        const uint64_t x0 = a.reqs.mask[(size_t)rep * rw + w] ^ t0[j * stride + w], x1 = a.strict.mask[(size_t)rep * rw + w] ^ t1[j * stride + w];
        d |= (((qj.defined >> kw) & 1u) ? x0 : 0ull) | (((qsj.defined >> kw) & 1u) ? x1 : 0ull);
      }
      if (l < nk) {   // one lane per key: minValues and bounds of the keys the set defines
        if (a.reqs.minv && ((qj.defined >> l) & 1u)) d |= (uint64_t)(uint32_t)(a.reqs.minv[(size_t)rep * nk + l] ^ m0[j * kstride + l]);
        if (a.strict.minv && ((qsj.defined >> l) & 1u)) d |= (uint64_t)(uint32_t)(a.strict.minv[(size_t)rep * nk + l] ^ m1[j * kstride + l]);
        if ((qj.has_gte >> l) & 1u) d |= (uint64_t)(a.reqs.gte[(size_t)rep * nk + l] ^ qj.gte[l]);
        if ((qj.has_lte >> l) & 1u) d |= (uint64_t)(a.reqs.lte[(size_t)rep * nk + l] ^ qj.lte[l]);
        if ((qsj.has_gte >> l) & 1u) d |= (uint64_t)(a.strict.gte[(size_t)rep * nk + l] ^ qsj.gte[l]);
        if ((qsj.has_lte >> l) & 1u) d |= (uint64_t)(a.strict.lte[(size_t)rep * nk + l] ^ qsj.lte[l]);
      }
      if (a.host_ports && l < 2) d |= a.host_ports[(size_t)rep * 2 + l] ^ a.host_ports[rowj * 2 + l];
      if (a.vol && l == 0) d |= a.vol[rep] ^ a.vol[rowj];
      if (a.topo_owned) for (int w = l; w < a.topo_words; w += 64) {
        d |= a.topo_owned[(size_t)rep * a.topo_words + w] ^ a.topo_owned[rowj * a.topo_words + w];
        d |= a.topo_selected[(size_t)rep * a.topo_words + w] ^ a.topo_selected[rowj * a.topo_words + w];
      }
      if (__ballot(d != 0 ? 1 : 0)) bad = true;
    }
  }
  if (bad) *a.collision = 1;
}
// ---- the classing kernel (pod equivalence classes: the HBM-streaming kernel of the path) ----
// One wavefront per 64 consecutive rows, and ONE round trip to HBM for everything the block reads: the 64 x req_words mask
// words of both requirement tables and the 64 x n_keys minValues of both are contiguous, so lane l takes 16-byte pieces
// l, l+64, ... (1 KiB per access, fully coalesced) — every access of the block is issued before the first value is used,
// 20 + 8 + 14 loads per lane in flight for a 1280-value dictionary — and drops them at [row][word] in LDS (odd row stride:
// the per-row reads that follow spread over the banks). Every lane then hashes its own row out of LDS and registers.
// The class table is one address per class for a million rows: it is touched once per DISTINCT hash of the wavefront
// (neighbouring pods mostly share their class), by the first lane that carries the hash, all those lanes at once: slot and
// representative in one round trip, the representative's row in another (row_diff_far). The other lanes take their leader's
// slot and have to equal ITS row (equality is transitive) — both rows are in LDS.
constexpr int kCoopMaskCh = 10;   // 16-byte mask accesses per lane, table and round: 64 rows x 20 words
constexpr int kCoopMinvCh = 4;    // 16-byte minValues accesses per lane, table and round: 64 rows x 16 keys
__device__ __forceinline__ uint4 coop_load_u64x2(const uint64_t* g, int e, int total) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (e + 2 <= total) v = *(const uint4*)(g + e);
  else if (e < total) { const uint64_t x = g[e]; v.x = (uint32_t)x; v.y = (uint32_t)(x >> 32); }
  return v;
}
__device__ __forceinline__ uint4 coop_load_i32x4(const int32_t* g, int e, int total) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (e + 4 <= total) v = *(const uint4*)(g + e);
  else {
    if (e < total) v.x = (uint32_t)g[e];
    if (e + 1 < total) v.y = (uint32_t)g[e + 1];
    if (e + 2 < total) v.z = (uint32_t)g[e + 2];
  }
  return v;
}
template <bool WHOLE = false>   // WHOLE: a piece is inside the table or past its end, never across it
__device__ __forceinline__ void coop_drop_u64x2(uint64_t* t, uint4 v, int e, int total, int rw, int stride, uint32_t magic) {
  if (WHOLE) {
    if (e < total) {
      const int r = (int)__umulhi((uint32_t)e, magic), r1 = (int)__umulhi((uint32_t)(e + 1), magic);
      t[r * stride + (e - r * rw)] = (uint64_t)v.x | ((uint64_t)v.y << 32);
      t[r1 * stride + (e + 1 - r1 * rw)] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    return;
  }
  if (e < total) { const int r = (int)__umulhi((uint32_t)e, magic); t[r * stride + (e - r * rw)] = (uint64_t)v.x | ((uint64_t)v.y << 32); }
  if (e + 1 < total) { const int r = (int)__umulhi((uint32_t)(e + 1), magic); t[r * stride + (e + 1 - r * rw)] = (uint64_t)v.z | ((uint64_t)v.w << 32); }
}
template <bool WHOLE = false>
__device__ __forceinline__ void coop_drop_i32x4(int32_t* t, uint4 v, int e, int total, int nk, int kstride, uint32_t kmagic) {
  const uint32_t x[4] = {v.x, v.y, v.z, v.w};
  if (WHOLE && e >= total) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) if (WHOLE || e + i < total) { const int r = (int)__umulhi((uint32_t)(e + i), kmagic); t[r * kstride + (e + i - r * nk)] = (int32_t)x[i]; }
}
namespace ks {
struct CoopStage {
  const uint64_t *g0, *g1; const int32_t *k0, *k1;
  uint64_t *t0, *t1; int32_t *m0, *m1;
  int total, ktotal, rw, stride, nk, kstride; uint32_t magic, kmagic; int l;
};
}
// FULL: 64 rows, so both totals are multiples of the access width and an access is either whole or past the end; an access
// past the end reads piece 0 instead (no branch around a load: a branch makes the compiler wait for every load before it).
// SAME: the strict table IS the requirement table (no pod of the batch has a preference: PodData.StrictRequirements ==
// Requirements, scheduler.go:217-229, and the flattener hands both under one pointer) — it is read and staged once.
template <bool FULL, bool MINV, bool SAME, int NRM>
__device__ __forceinline__ void coop_stage(const ks::CoopStage& s, const ks::RowArgs& a, int rowc, ks::ReqRef& q, ks::ReqRef& qs, int64_t (&rq)[NRM], uint64_t& tol) {
  uint4 A[kCoopMaskCh], B[SAME ? 1 : kCoopMaskCh], KA[kCoopMinvCh], KB[SAME ? 1 : kCoopMinvCh];
  const int l = s.l, nr = a.n_res;
#pragma unroll
  for (int i = 0; i < kCoopMaskCh; ++i) {
    const int e = 2 * (l + 64 * i);
    if (FULL) { const int x = e < s.total ? e : 0; A[i] = *(const uint4*)(s.g0 + x); if (!SAME) B[i] = *(const uint4*)(s.g1 + x); }
    else { A[i] = coop_load_u64x2(s.g0, e, s.total); if (!SAME) B[i] = coop_load_u64x2(s.g1, e, s.total); }
  }
  if (MINV) {
#pragma unroll
    for (int i = 0; i < kCoopMinvCh; ++i) {
      const int e = 4 * (l + 64 * i);
      if (FULL) { const int x = e < s.ktotal ? e : 0; KA[i] = *(const uint4*)(s.k0 + x); if (!SAME) KB[i] = *(const uint4*)(s.k1 + x); }
      else { KA[i] = coop_load_i32x4(s.k0, e, s.ktotal); if (!SAME) KB[i] = coop_load_i32x4(s.k1, e, s.ktotal); }
    }
  }
  q.defined = a.reqs.defined[rowc]; q.complement = a.reqs.complement[rowc]; q.has_gte = a.reqs.has_gte[rowc]; q.has_lte = a.reqs.has_lte[rowc];
  if (SAME) { qs.defined = q.defined; qs.complement = q.complement; qs.has_gte = q.has_gte; qs.has_lte = q.has_lte; }
  else { qs.defined = a.strict.defined[rowc]; qs.complement = a.strict.complement[rowc]; qs.has_gte = a.strict.has_gte[rowc]; qs.has_lte = a.strict.has_lte[rowc]; }
#pragma unroll
  for (int r = 0; r < NRM; ++r) rq[r] = a.requests[(size_t)(r < nr ? r : 0) * a.n_rows + rowc];
  tol = a.tolerates[rowc];
#pragma unroll
  for (int i = 0; i < kCoopMaskCh; ++i) { const int e = 2 * (l + 64 * i); coop_drop_u64x2<FULL>(s.t0, A[i], e, s.total, s.rw, s.stride, s.magic); if (!SAME) coop_drop_u64x2<FULL>(s.t1, B[i], e, s.total, s.rw, s.stride, s.magic); }
  if (MINV) {
#pragma unroll
    for (int i = 0; i < kCoopMinvCh; ++i) { const int e = 4 * (l + 64 * i); coop_drop_i32x4<FULL>(s.m0, KA[i], e, s.ktotal, s.nk, s.kstride, s.kmagic); if (!SAME) coop_drop_i32x4<FULL>(s.m1, KB[i], e, s.ktotal, s.nk, s.kstride, s.kmagic); }
  }
  for (int base = 2 * 64 * kCoopMaskCh; base < s.total; base += 2 * 64 * kCoopMaskCh) {   // dictionaries beyond one round
#pragma unroll
    for (int i = 0; i < kCoopMaskCh; ++i) { const int e = base + 2 * (l + 64 * i); A[i] = coop_load_u64x2(s.g0, e, s.total); if (!SAME) B[i] = coop_load_u64x2(s.g1, e, s.total); }
#pragma unroll
    for (int i = 0; i < kCoopMaskCh; ++i) { const int e = base + 2 * (l + 64 * i); coop_drop_u64x2(s.t0, A[i], e, s.total, s.rw, s.stride, s.magic); if (!SAME) coop_drop_u64x2(s.t1, B[i], e, s.total, s.rw, s.stride, s.magic); }
  }
  if (MINV) for (int base = 4 * 64 * kCoopMinvCh; base < s.ktotal; base += 4 * 64 * kCoopMinvCh) {
#pragma unroll
    for (int i = 0; i < kCoopMinvCh; ++i) { const int e = base + 4 * (l + 64 * i); KA[i] = coop_load_i32x4(s.k0, e, s.ktotal); if (!SAME) KB[i] = coop_load_i32x4(s.k1, e, s.ktotal); }
#pragma unroll
    for (int i = 0; i < kCoopMinvCh; ++i) { const int e = base + 4 * (l + 64 * i); coop_drop_i32x4(s.m0, KA[i], e, s.ktotal, s.nk, s.kstride, s.kmagic); if (!SAME) coop_drop_i32x4(s.m1, KB[i], e, s.ktotal, s.nk, s.kstride, s.kmagic); }
  }
}
// MINV: the rows carry minValues (pod rows never do when they come from the reference's PodData — minValues belong to NodePool
// requirements — so the tables are absent and neither streamed nor staged)
template <bool MINV, bool SAME, int NRM>
__device__ __forceinline__ void row_hash_coop2_body(uint64_t* coop_lds, int n, const ks::RowArgs& a, int rw, uint32_t magic, uint32_t kmagic, int rpb) {
  const int l = (int)threadIdx.x;
  const int row0 = (int)blockIdx.x * rpb;   // rpb rows per block: 64 (60, a multiple of 4, only through the A/B switch of the launcher)
  const int rows = n - row0 < rpb ? n - row0 : rpb;
  const int stride = rw | 1;
  const int total = rows * rw;
  const int nk = a.dict.n_keys, kstride = nk | 1, ktotal = rows * nk;
  const int nr = a.n_res;
  uint64_t* t0 = coop_lds;
  uint64_t* t1 = SAME ? t0 : coop_lds + rpb * stride;
  int32_t* m0 = (int32_t*)(coop_lds + (SAME ? 1 : 2) * rpb * stride);
  int32_t* m1 = SAME ? m0 : m0 + rpb * kstride;
  const uint64_t* g0 = a.reqs.mask + (size_t)row0 * rw;
  const uint64_t* g1 = a.strict.mask + (size_t)row0 * rw;
  const int32_t* k0 = MINV ? a.reqs.minv + (size_t)row0 * nk : nullptr;   // the launcher: both minValues tables or neither, every other optional table exists
  const int32_t* k1 = MINV ? a.strict.minv + (size_t)row0 * nk : nullptr;
  const int row = row0 + l;
  const bool live = l < rows;
  const int rowc = live ? row : n - 1;
  // ---- every HBM access of the block, then [row][word] in LDS ----
  ks::ReqRef q, qs;
  int64_t rq[NRM];   // NRM = 4 when the problem has at most four resource dimensions (the launcher knows)
  uint64_t tol;
  const ks::CoopStage st{g0, g1, k0, k1, t0, t1, m0, m1, total, ktotal, rw, stride, nk, kstride, magic, kmagic, l};
  if (rows == rpb) coop_stage<true, MINV, SAME, NRM>(st, a, rowc, q, qs, rq, tol);     // branch-free: no wait is placed before the last load is out
  else coop_stage<false, MINV, SAME, NRM>(st, a, rowc, q, qs, rq, tol);               // the last block of the table
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- the row's hash, out of LDS and registers ----
  q.mask = t0 + (size_t)l * stride; qs.mask = t1 + (size_t)l * stride;
  q.minv = MINV ? m0 + (size_t)l * kstride : nullptr; qs.minv = MINV ? m1 + (size_t)l * kstride : nullptr;
  q.gte = a.reqs.gte ? a.reqs.gte + (size_t)rowc * nk : nullptr; q.lte = a.reqs.lte ? a.reqs.lte + (size_t)rowc * nk : nullptr;
  qs.gte = a.strict.gte ? a.strict.gte + (size_t)rowc * nk : nullptr; qs.lte = a.strict.lte ? a.strict.lte + (size_t)rowc * nk : nullptr;
  auto req_at = [&](int r) -> int64_t { return rq[r]; };   // every caller unrolls over r: register indices
  const uint64_t h = live ? ks::row_hash_kept(a, ks::row_hash_value_with<SAME, NRM>(row, a, q, qs, req_at, tol)) : 0ull;
  // ---- the first lane of every distinct hash ----
  int lead = l;
  uint64_t todo = __ballot(live ? 1 : 0);
  while (todo) {
    const int j = __builtin_ctzll(todo);
    const uint32_t hlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h, j), hhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(h >> 32), j);
    const bool mine = live && h == ((uint64_t)hlo | ((uint64_t)hhi << 32));
    todo &= ~__ballot(mine ? 1 : 0);
    if (mine) lead = j;
  }
  const bool leader = live && lead == l;
  uint32_t slot = 0, rep = 0xFFFFFFFFu;
  if (leader) rep = ks::row_table_insert(row, a, h, &slot);
  slot = (uint32_t)__shfl((int)slot, lead, 64);
  if (live) a.row_slot[row] = slot;
  uint64_t d = 0;
  if (leader && rep != 0xFFFFFFFFu) d = ks::row_diff_far<SAME, MINV, NRM>(row, a, rep, q, qs, req_at, tol);
  // ---- a follower equals its leader: flags and requests by shuffle, mask words and minValues LDS to LDS ----
  ks::ReqRef qj, qsj;
  qj.defined = (uint32_t)__shfl((int)q.defined, lead, 64); qj.complement = (uint32_t)__shfl((int)q.complement, lead, 64);
  qj.has_gte = (uint32_t)__shfl((int)q.has_gte, lead, 64); qj.has_lte = (uint32_t)__shfl((int)q.has_lte, lead, 64);
  qsj.defined = (uint32_t)__shfl((int)qs.defined, lead, 64); qsj.complement = (uint32_t)__shfl((int)qs.complement, lead, 64);
  qsj.has_gte = (uint32_t)__shfl((int)qs.has_gte, lead, 64); qsj.has_lte = (uint32_t)__shfl((int)qs.has_lte, lead, 64);
  uint64_t dq = (uint64_t)__shfl((unsigned long long)tol, lead, 64) ^ tol;
#pragma unroll
  for (int r = 0; r < NRM; ++r) if (r < nr) dq |= (uint64_t)__shfl((unsigned long long)rq[r], lead, 64) ^ (uint64_t)rq[r];
  if (live && !leader) {
    const size_t rowj = (size_t)(row0 + lead);
    qj.mask = t0 + (size_t)lead * stride; qsj.mask = t1 + (size_t)lead * stride;
    qj.minv = MINV ? m0 + (size_t)lead * kstride : nullptr; qsj.minv = MINV ? m1 + (size_t)lead * kstride : nullptr;
    qj.gte = a.reqs.gte ? a.reqs.gte + rowj * nk : nullptr; qj.lte = a.reqs.lte ? a.reqs.lte + rowj * nk : nullptr;
    qsj.gte = a.strict.gte ? a.strict.gte + rowj * nk : nullptr; qsj.lte = a.strict.lte ? a.strict.lte + rowj * nk : nullptr;
    d = dq | ks::reqset_diff(a.dict, q, qj);
    if (!SAME) d |= ks::reqset_diff(a.dict, qs, qsj);
    if (a.host_ports) d |= (a.host_ports[(size_t)row * 2] ^ a.host_ports[rowj * 2]) | (a.host_ports[(size_t)row * 2 + 1] ^ a.host_ports[rowj * 2 + 1]);
        if (a.vol) d |= a.vol[row] ^ a.vol[rowj];
    if (a.topo_owned) for (int w = 0; w < a.topo_words; ++w) {
      d |= a.topo_owned[(size_t)row * a.topo_words + w] ^ a.topo_owned[rowj * a.topo_words + w];
      d |= a.topo_selected[(size_t)row * a.topo_words + w] ^ a.topo_selected[rowj * a.topo_words + w];
    }
  }
  if (d) *a.collision = 1;
}
template <bool MINV, bool SAME = false, int NRM = 8>
__global__ void __launch_bounds__(64) ksolve_row_hash_coop2(int n, ks::RowArgs a, int rw, uint32_t magic, uint32_t kmagic, int rpb) {
  extern __shared__ __attribute__((aligned(16))) uint64_t coop_lds[];
  row_hash_coop2_body<MINV, SAME, NRM>(coop_lds, n, a, rw, magic, kmagic, rpb);
}
// (A build of the one-table form at four wavefronts per SIMD — amdgpu_waves_per_eu(4,4), 128 VGPRs instead of the 149 it takes —
// measured 157 us against 79.6 us: the spills cost more than the fourth wavefront hides; profiles/README.md.)
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
// every pod class against every pristine node of a resident cluster (kernels.h node_dead0_body): one wavefront per 64 nodes x 32 classes
__global__ void __launch_bounds__(64) ksolve_node_dead0(ks::NodeDeadArgs a) { ks::node_dead0_body<ks::Wave>((int)blockIdx.x, (int)blockIdx.y, a); }
__global__ void ksolve_claim_gather(int n, ks::ClaimGatherArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::claim_gather_body(i, a);
}
// a sweep's workspace records from their 48-byte descriptors (kernels.h sweep_item_fill), one thread per probe
__global__ void ksolve_sweep_items(int n, ks::SweepItemArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::sweep_items_body(i, a);
}
__global__ void ksolve_fast_queue(int n, ks::FastQueueArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::fast_queue_body(i, a);
}
__global__ void ksolve_fast_overlap(int nc, ks::FastQueueArgs a) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nc) ks::fast_overlap_body(c, nc, a);
}
__global__ void ksolve_fast_mark(int n, ks::FastQueueArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ks::fast_mark_body(i, a);
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
static void be_launch_row_hash(ksolve_handle* h, int n, const ks::RowArgs& a) {
  const int rw = a.dict.req_words;
  const int nk = a.dict.n_keys;
  // A/B switch for measurements (scripts/gpu_r3_classing.sh): "coop1" = the previous wave-cooperative kernel, "plain" = one thread per row
#ifdef KSOLVE_TEST_HOOKS
  const char* variant = getenv("KSOLVE_ROWHASH_KERNEL");
  const bool no_shared_strict = getenv("KSOLVE_TEST_NO_SHARED_STRICT") != nullptr;
#else
  const char* variant = nullptr;   // the product binary reads no test / measurement switch (tests/emu/libksolve_hooks.so is the build that does)
  const bool no_shared_strict = false;
#endif
  const bool aligned = (((uintptr_t)a.reqs.mask | (uintptr_t)a.strict.mask | (uintptr_t)a.reqs.minv | (uintptr_t)a.strict.minv) & 15) == 0;
  // floor(e / rw) = umulhi(e, magic) for every e < 64 * rw (e * rw < 2^32)
  const uint32_t magic = rw >= 1 ? (uint32_t)((0x100000000ull + (uint64_t)rw - 1) / (uint64_t)rw) : 0;
  const uint32_t kmagic = nk >= 1 ? (uint32_t)((0x100000000ull + (uint64_t)nk - 1) / (uint64_t)nk) : 0;
  const bool minv = a.reqs.minv && a.strict.minv, no_minv = !a.reqs.minv && !a.strict.minv;
  // 64 rows per block. 60 rows (4 idle lanes) would let an eighth block fit the CU's 160 KiB of LDS at configs[1]'s dictionary;
  // measured (profiles/round2/classing_ab3.log): 153 us against 150.5 us — the idle lanes cost what the wavefront buys.
  // one table to stage when the strict requirements ARE the requirements (same device table): half the LDS, half the loads
  const bool same = a.strict.mask == a.reqs.mask && a.strict.minv == a.reqs.minv && a.strict.defined == a.reqs.defined && !no_shared_strict;
  auto lds_for = [&](int rpb) { return (size_t)(same ? 1 : 2) * rpb * (size_t)(rw | 1) * 8 + (minv ? (size_t)(same ? 1 : 2) * rpb * (size_t)(nk | 1) * 4 : 0); };
  int rpb = 64;
#ifdef KSOLVE_TEST_HOOKS
  if (const char* r = getenv("KSOLVE_TEST_ROWS_PER_BLOCK")) { const int v = atoi(r); if (v == 60 || v == 64) rpb = v; }   // A/B switch of tests/tools/gpu_classing_ab.py
#endif
  size_t lds2 = lds_for(rpb);
#ifdef KSOLVE_TEST_HOOKS
  if (const char* pad = getenv("KSOLVE_TEST_LDS_PAD")) lds2 += (size_t)atoi(pad);   // occupancy probe of tests/tools/gpu_classing_ab.py: fewer wavefronts per CU
#endif
  const size_t lds1 = (size_t)2 * 64 * (size_t)(rw | 1) * 8 + (size_t)2 * 64 * (size_t)(nk | 1) * 4 + 8 + (size_t)64 * (size_t)(a.n_res + 1) * 8;
  const bool plain = variant && !strcmp(variant, "plain");
  const bool coop1 = variant && !strcmp(variant, "coop1");
  const bool tables = (minv || no_minv) && a.reqs.has_gte && a.reqs.has_lte && a.strict.has_gte && a.strict.has_lte;
  if (!plain && !coop1 && rw >= 1 && nk >= 1 && a.n_res >= 1 && a.n_res <= 8 && aligned && tables && lds2 <= 64 * 1024) {
    const dim3 grid((unsigned)((n + rpb - 1) / rpb));
    const bool nr4 = a.n_res <= 4;   // four request registers instead of eight (cpu, memory, pods, ephemeral-storage: the usual problem)
    if (minv && same) hipLaunchKernelGGL((ksolve_row_hash_coop2<true, true>), grid, dim3(64), lds2, HB(h)->stream, n, a, rw, magic, kmagic, rpb);
    else if (minv) hipLaunchKernelGGL((ksolve_row_hash_coop2<true, false>), grid, dim3(64), lds2, HB(h)->stream, n, a, rw, magic, kmagic, rpb);
    else if (same && nr4) hipLaunchKernelGGL((ksolve_row_hash_coop2<false, true, 4>), grid, dim3(64), lds2, HB(h)->stream, n, a, rw, magic, kmagic, rpb);
    else if (same) hipLaunchKernelGGL((ksolve_row_hash_coop2<false, true>), grid, dim3(64), lds2, HB(h)->stream, n, a, rw, magic, kmagic, rpb);
    else if (nr4) hipLaunchKernelGGL((ksolve_row_hash_coop2<false, false, 4>), grid, dim3(64), lds2, HB(h)->stream, n, a, rw, magic, kmagic, rpb);
    else hipLaunchKernelGGL((ksolve_row_hash_coop2<false, false>), grid, dim3(64), lds2, HB(h)->stream, n, a, rw, magic, kmagic, rpb);
  } else if (!plain && rw >= 1 && nk >= 1 && lds1 <= 64 * 1024) {
    hipLaunchKernelGGL(ksolve_row_hash_coop, dim3((unsigned)((n + 63) / 64)), dim3(64), lds1, HB(h)->stream, n, a, rw, magic, kmagic);
  } else {
    hipLaunchKernelGGL(ksolve_row_hash, grid_for(n), dim3(256), 0, HB(h)->stream, n, a);
  }
}
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

static int be_device_of(const ksolve_handle* h) { return HB(h)->device; }
static void be_free(ksolve_handle* h, void* p) {
  if (!p) return;
  hip_check(h, hipStreamSynchronize(HB(h)->stream), "hipStreamSynchronize");
  auto it = std::find(h->allocations.begin(), h->allocations.end(), p);
  if (it != h->allocations.end()) h->allocations.erase(it);
  (void)hipFree(p);
}
static void be_launch_node_dead0(ksolve_handle* h, int n_blocks, const ks::NodeDeadArgs& a) {
  hipLaunchKernelGGL(ksolve_node_dead0, dim3((unsigned)n_blocks, (unsigned)((a.n_classes + ks::kDead0Classes - 1) / ks::kDead0Classes)), dim3(64), 0, HB(h)->stream, a);
  hip_check(h, hipGetLastError(), "ksolve_node_dead0 launch");
}
static void be_launch_claim_gather(ksolve_handle* h, int n, const ks::ClaimGatherArgs& a) {
  hipLaunchKernelGGL(ksolve_claim_gather, grid_for(n), dim3(256), 0, HB(h)->stream, n, a);
  hip_check(h, hipGetLastError(), "ksolve_claim_gather launch");
}
static void be_launch_sweep_items(ksolve_handle* h, int n, const ks::SweepItemArgs& a) {
  hipLaunchKernelGGL(ksolve_sweep_items, grid_for(n), dim3(256), 0, HB(h)->stream, n, a);
  hip_check(h, hipGetLastError(), "ksolve_sweep_items launch");
}
static void be_launch_pack_sweep(ksolve_handle* h, const ks::ProblemView* d_pv, ks::Workspace* d_items, int n, const ks::LdsPlan& plan, const uint32_t* d_order, uint32_t* d_next) {
  if (n <= 0) return;
  HipBackend* b = HB(h);
  const int lds1 = ((plan.total_bytes + 15) & ~15) + ks::kSweepLdsExtra;   // + the view and one workspace record (ksolve_pack_sweep)
  if (!hip_check(h, hipFuncSetAttribute((const void*)ksolve_pack_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, lds1), "hipFuncSetAttribute(LDS)")) return;
  if (plan.waves == 4) {
    // the compact form: as many workgroups as the chip holds at once (two per CU by registers, fewer when their LDS is large), each
    // wavefront striding over the probes — the shared tables and the template prefilter are paid once per workgroup
    const int lds4 = ((plan.total_bytes + 15) & ~15) + ks::kSweepLdsExtra;   // + the view and the workspace records (ksolve_pack_sweep4)
    if (!hip_check(h, hipFuncSetAttribute((const void*)ksolve_pack_sweep4, hipFuncAttributeMaxDynamicSharedMemorySize, lds4), "hipFuncSetAttribute(LDS)")) return;
    int per_cu = (160 * 1024) / (lds4 + 256);
    per_cu = per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu;
    const int resident = b->n_cus * per_cu, want = (n + 3) / 4;
    const int grid4 = want < resident ? want : resident;
    hip_check(h, hipEventRecord(b->ev0[ksi::T_PACK], b->stream), "hipEventRecord");
    hipLaunchKernelGGL(ksolve_pack_sweep4, dim3((unsigned)grid4), dim3(256), (size_t)lds4, b->stream, d_pv, d_items, n, plan, d_order, d_next);
    hip_check(h, hipGetLastError(), "ksolve_pack_sweep4 launch");
    hip_check(h, hipEventRecord(b->ev1[ksi::T_PACK], b->stream), "hipEventRecord");
    hip_check(h, hipEventSynchronize(b->ev1[ksi::T_PACK]), "hipEventSynchronize");
    float ms4 = 0;
    if (hipEventElapsedTime(&ms4, b->ev0[ksi::T_PACK], b->ev1[ksi::T_PACK]) == hipSuccess) h->timers.ms[ksi::T_PACK] = ms4;
    return;
  }
  const int grid = n < 8192 ? n : 8192;
  hip_check(h, hipEventRecord(b->ev0[ksi::T_PACK], b->stream), "hipEventRecord");
  hipLaunchKernelGGL(ksolve_pack_sweep, dim3((unsigned)grid), dim3(64), (size_t)lds1, b->stream, d_pv, d_items, n, plan);
  hip_check(h, hipGetLastError(), "ksolve_pack_sweep launch");
  hip_check(h, hipEventRecord(b->ev1[ksi::T_PACK], b->stream), "hipEventRecord");
  hip_check(h, hipEventSynchronize(b->ev1[ksi::T_PACK]), "hipEventSynchronize");
  float ms = 0;
  if (hipEventElapsedTime(&ms, b->ev0[ksi::T_PACK], b->ev1[ksi::T_PACK]) == hipSuccess) h->timers.ms[ksi::T_PACK] = ms;
}
typedef void (*ksolve_pack_fast_fn)(const ks::FastArgs*);
static ksolve_pack_fast_fn pack_fast_kernel(int plan, int rows) {
  if (rows == 1) return plan == 2 ? ksolve_pack_fast_g2r1 : plan == 1 ? ksolve_pack_fast_g1r1 : ksolve_pack_fast_g0r1;
  return plan == 2 ? ksolve_pack_fast_g2r4 : plan == 1 ? ksolve_pack_fast_g1r4 : ksolve_pack_fast_g0r4;
}
static void be_launch_pack_fast(ksolve_handle* h) {
  const int lds_bytes = h->fw.plan.total_bytes;
  const bool two = h->fw.plan.helper != 0;   // (plan 0, one row of class slots: placer + refresher)
  const ksolve_pack_fast_fn fn = two ? ksolve_pack_fast2 : pack_fast_kernel(h->fw.plan.global_state, h->fw.plan.rows);
  if (!hip_check(h, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)")) return;
  ks::FastArgs a{h->pv, h->ws, h->fw};
  be_h2d(h, h->d_fast_args, &a, sizeof(a));
  hipLaunchKernelGGL(fn, dim3(1), dim3(two ? 256 : 64), (size_t)lds_bytes, HB(h)->stream, (const ks::FastArgs*)h->d_fast_args);
  hip_check(h, hipGetLastError(), "ksolve_pack_fast launch");
}
static void be_launch_pack_topo(ksolve_handle* h) {
  const int lds_bytes = h->tw.plan.total_bytes;
  if (!hip_check(h, hipFuncSetAttribute((const void*)ksolve_pack_topo, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "hipFuncSetAttribute(LDS)")) return;
  ks::TopoArgs a{h->pv, h->ws, h->fw, h->tw};
  be_h2d(h, h->d_topo_args, &a, sizeof(a));
  hipLaunchKernelGGL(ksolve_pack_topo, dim3(1), dim3(64), (size_t)lds_bytes, HB(h)->stream, (const ks::TopoArgs*)h->d_topo_args);
  hip_check(h, hipGetLastError(), "ksolve_pack_topo launch");
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
  return ks::FastQueueArgs{h->pv.sorted_pods, h->pv.row_class, h->fw.q_class, h->fw.q_claim, h->fw.q_cnt, h->ws.assign, h->ws.slot, h->fw.cls_first, h->fw.cls_last, h->fw.max_active};
}
static void be_launch_fast_records(ksolve_handle* h, int n_claims) {
  ks::FastRecordArgs a{h->pv, h->ws, h->fw};
  hipLaunchKernelGGL(ksolve_fast_records, dim3((unsigned)n_claims), dim3(64), 0, HB(h)->stream, a);
  hip_check(h, hipGetLastError(), "ksolve_fast_records launch");
  const int n = (int)h->n_pods;
  hipLaunchKernelGGL(ksolve_fast_scatter, grid_for(n), dim3(256), 0, HB(h)->stream, n, fast_queue_args(h));
  hip_check(h, hipGetLastError(), "ksolve_fast_scatter launch");
}
static void be_launch_fast_queue(ksolve_handle* h, bool count_live) {
  const int n = (int)h->n_pods;
  hipLaunchKernelGGL(ksolve_fast_queue, grid_for(n), dim3(256), 0, HB(h)->stream, n, fast_queue_args(h));
  hip_check(h, hipGetLastError(), "ksolve_fast_queue launch");
  const int nc = (int)h->n_classes;
  if (count_live && h->fw.enabled && nc > 64 && nc <= 32768) {   // (the spread engine has no class slots to count)
    hipLaunchKernelGGL(ksolve_fast_overlap, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, HB(h)->stream, nc, fast_queue_args(h));
    hip_check(h, hipGetLastError(), "ksolve_fast_overlap launch");
  }
  hipLaunchKernelGGL(ksolve_fast_mark, grid_for(n), dim3(256), 0, HB(h)->stream, n, fast_queue_args(h));
  hip_check(h, hipGetLastError(), "ksolve_fast_mark launch");
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
  for (size_t i = 0; i < g.size(); ++i) { items[i].pv = g[i]->pv; items[i].ws = g[i]->ws; lds_bytes = std::max(lds_bytes, ((g[i]->pv.lds.total_bytes + 15) & ~15) + ks::kSweepLdsExtra); }   // + the block's copy of its view and workspace record
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
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) == hipSuccess && cus > 0) b->n_cus = cus; }
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
  b->device = HB(base)->device; b->n_cus = HB(base)->n_cus;
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
ksolve_status ksolve_sweep(ksolve_handle* base, const ksolve_sweep_desc* desc, ksolve_sweep_results* out) {
  if (!base || !base->backend || !out) return KSOLVE_ERR_INVALID;
  if (hipSetDevice(HB(base)->device) != hipSuccess) return KSOLVE_ERR_DEVICE;
  return ksi::sweep(base, desc, out);
}
ksolve_status ksolve_sweep_replicas(ksolve_handle** bases, uint32_t n_bases, const ksolve_sweep_desc* desc, ksolve_sweep_results* out) {
  if (!bases || !n_bases || !out) return KSOLVE_ERR_INVALID;
  for (uint32_t i = 0; i < n_bases; ++i) if (!bases[i] || !bases[i]->backend) return KSOLVE_ERR_INVALID;
  if (hipSetDevice(HB(bases[0])->device) != hipSuccess) return KSOLVE_ERR_DEVICE;
  return ksi::sweep_replicas(bases, n_bases, desc, out);
}
void ksolve_sweep_results_free(ksolve_sweep_results* r) { if (r && r->impl) { delete (ksi::SweepImpl*)r->impl; r->impl = nullptr; } }
ksolve_status ksolve_solve_batch(ksolve_handle** hs, uint32_t n, ksolve_results* outs) {
  if (!hs || !outs || n == 0) return KSOLVE_ERR_INVALID;
  for (uint32_t i = 0; i < n; ++i) if (!hs[i] || !hs[i]->backend) return KSOLVE_ERR_INVALID;
  if (hipSetDevice(HB(hs[0])->device) != hipSuccess) return KSOLVE_ERR_DEVICE;
  return ksi::solve_batch(hs, n, outs);
}
ksolve_status ksolve_packing_vector(const ksolve_handle* h, const ksolve_results* r, double* count, double* cost) {
  if (!h || !r || !count || !cost) return KSOLVE_ERR_INVALID;
  return ksi::packing_vector(h, r->claims, count, cost);
}
ksolve_status ksolve_packing_vector_sum(ksolve_handle* const* hs, const ksolve_results* rs, uint32_t n, double* count, double* cost) {
  if (!hs || !rs || !count || !cost || n == 0) return KSOLVE_ERR_INVALID;
  return ksi::packing_vector_sum(hs, rs, n, count, cost);
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
    if (b->stage) (void)hipHostFree(b->stage);
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
  if (n == "row_hash") return h->timers.ms[ksi::T_ROWHASH];
  if (n == "sort") return h->timers.ms[ksi::T_SORT];
  if (n == "it_index") return h->timers.ms[ksi::T_INDEX];
  return -1;
}

}  // extern "C"
