// TEST INFRASTRUCTURE ONLY — host emulation of the device solver.
//
// Compiles the product's device algorithm (karpenter_amd/csrc/{engine,pdq_emul,reqalg,kernels}.h) for the host with
// KSOLVE_HOST_EMULATION, where a wavefront is a loop over 64 lanes and a kernel launch is a loop over its grid. It
// exists so that `pytest -m "not gpu"` can fuzz the device algorithm against the oracle on machines without a GPU.
// It is built only by tests/ (into tests/emu/), is never built by __graft_entry__.build() as part of the product, and
// the product's Python host (karpenter_amd/scheduling.py) refuses to load it unless a test passes it explicitly.
#define KSOLVE_HOST_EMULATION 1
#define KSOLVE_TEST_HOOKS 1   // the test switches (KSOLVE_TEST_*) exist in the test builds only
#include <algorithm>
#include "../../karpenter_amd/csrc/ksolve_impl.h"
#include "../../karpenter_amd/csrc/topo_engine.h"

struct EmuBackend { std::chrono::steady_clock::time_point t0[8]; };

static void* be_alloc(ksolve_handle* h, size_t bytes) { void* p = calloc(1, bytes ? bytes : 1); h->allocations.push_back(p); return p; }
// KSOLVE_TEST_ARENA_LIMIT_MB: an emulated device that refuses a sweep arena above that size (tests of the halving retry)
static void* be_try_alloc(ksolve_handle* h, size_t bytes) {
  if (const char* e = getenv("KSOLVE_TEST_ARENA_LIMIT_MB")) if (bytes > ((size_t)atoi(e) << 20)) return nullptr;
  return be_alloc(h, bytes);
}
static void be_h2d(ksolve_handle*, void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
static void be_d2h(ksolve_handle*, void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
static void be_fill(ksolve_handle*, void* dst, int byte, size_t bytes) { memset(dst, byte, bytes); }
static void be_sync(ksolve_handle*) {}
static void* be_stage(ksolve_handle* h, size_t bytes) {   // the device build: page-locked memory kept by the handle
  static thread_local std::vector<char> stage;
  (void)h;
  if (stage.size() < bytes) stage.resize(bytes + bytes / 4 + 4096);
  return stage.data();
}
static void be_thread_init(ksolve_handle*) {}
static bool be_ok(ksolve_handle*) { return true; }
static void be_tic(ksolve_handle* h, int slot) { ((EmuBackend*)h->backend)->t0[slot] = std::chrono::steady_clock::now(); }
static void be_range_drop(ksolve_handle*) {}
static void be_toc(ksolve_handle* h, int slot) {
  h->timers.ms[slot] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ((EmuBackend*)h->backend)->t0[slot]).count();
}
static void be_launch_it_index(ksolve_handle*, int n, const ks::ItIndexArgs& a) { for (int i = 0; i < n; ++i) ks::it_index_body(i, a); }
static void be_launch_row_hash(ksolve_handle*, int n, const ks::RowArgs& a) { for (int i = 0; i < n; ++i) ks::row_hash_body(i, a); }
static void be_launch_row_class(ksolve_handle*, int n, const ks::RowArgs& a) { for (int i = 0; i < n; ++i) ks::row_class_body(i, a); }
static void be_launch_class_gather(ksolve_handle*, int n, const ks::RowArgs& a) { for (int i = 0; i < n; ++i) ks::class_gather_body(i, a); }
static void be_launch_finalize(ksolve_handle*, int n, const ks::FinalizeArgs& a) { for (int i = 0; i < n; ++i) ks::finalize_body(i, a); }
static void be_sort_pods(ksolve_handle* h) {
  // same LSD structure as the device path: five stable passes over 64-bit keys
  const int n = (int)h->n_pods;
  uint32_t* idx = h->d_idx_a;
  for (int i = 0; i < n; ++i) idx[i] = (uint32_t)i;
  for (int pass = 0; pass < 5; ++pass) {
    ks::SortKeyArgs a = h->sort_args;
    a.idx_in = idx; a.key_out = h->d_key_a; a.pass = pass;
    for (int i = 0; i < n; ++i) ks::sort_key_body(i, a);
    std::vector<uint32_t> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = (uint32_t)i;
    const uint64_t* key = h->d_key_a;
    std::stable_sort(perm.begin(), perm.end(), [key](uint32_t x, uint32_t y) { return key[x] < key[y]; });
    std::vector<uint32_t> out(n);
    for (int i = 0; i < n; ++i) out[i] = idx[perm[i]];
    memcpy(idx, out.data(), (size_t)n * 4);
  }
  h->pv.sorted_pods = idx;
}
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
// AddressSanitizer build (scripts/asan_check.sh): the LDS segment is one allocation, so an overflow from one LDS table
// into the next would go unnoticed — spread the tables apart and poison the gaps.
static ks::LdsPlan guarded_plan(const ks::LdsPlan& in, std::vector<std::pair<int, int>>& gaps) {
  ks::LdsPlan p = in;
  int* offs[] = {&p.off_alloc, &p.off_avail, &p.off_kv, &p.off_keymask, &p.off_allocok, &p.off_kvslot, &p.off_tmpl, &p.off_tmplcold, &p.off_order,
                 &p.off_closed, &p.off_stage, &p.off_cache, &p.off_scratch, &p.off_dgov, &p.off_dgits, &p.off_topo};
  if (!in.topo_bytes) p.off_topo = p.off_cache;   // (not planned: rides with a region that is)
  std::vector<int> starts;
  for (int* o : offs) starts.push_back(*o);
  std::sort(starts.begin(), starts.end());
  starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
  const int pad = 64;
  for (int* o : offs) { int rank = (int)(std::lower_bound(starts.begin(), starts.end(), *o) - starts.begin()); *o += pad * rank; }
  for (size_t i = 1; i < starts.size(); ++i) gaps.push_back({starts[i] + pad * (int)(i - 1), pad});   // the gap sits right before region i
  p.total_bytes = in.total_bytes + pad * (int)starts.size();
  gaps.push_back({p.total_bytes - pad, pad});
  return p;
}
#endif
static void be_launch_pack(ksolve_handle* h) {
#if defined(__SANITIZE_ADDRESS__)
  std::vector<std::pair<int, int>> gaps;
  ks::ProblemView pv = h->pv;
  pv.lds = guarded_plan(h->pv.lds, gaps);
  std::vector<char> lds((size_t)pv.lds.total_bytes + 64, (char)0xA5);   // LDS is not zeroed at kernel start on the device either
  for (auto& g : gaps) __asan_poison_memory_region(lds.data() + g.first, (size_t)g.second);
#else
  const ks::ProblemView& pv = h->pv;
  // stands in for the CU's LDS segment; filled with garbage because the device does not zero LDS at kernel start: an
  // engine that relied on zeroed LDS would pass here and fail on the GPU
  std::vector<char> lds((size_t)pv.lds.total_bytes + 64, (char)0xA5);
#endif
  ks::LdsTables tables;
  tables.bind(lds.data(), pv.lds);
  if (pv.lds.topo_bytes) tables.topo = lds.data() + pv.lds.off_topo;   // (one problem per launch: the topology groups' descriptors and small state in LDS)
  if (pv.big) { ks::Engine<ks::Wave, true, true> eng(pv, h->ws, tables); eng.solve(); }
  else if (pv.lite) { ks::Engine<ks::Wave, false> eng(pv, h->ws, tables); eng.solve(); }
  else { ks::Engine<ks::Wave, true> eng(pv, h->ws, tables); eng.solve(); }
#if defined(__SANITIZE_ADDRESS__)
  for (auto& g : gaps) __asan_unpoison_memory_region(lds.data() + g.first, (size_t)g.second);
#endif
}
static void be_launch_pack_fast(ksolve_handle* h) {
  std::vector<char> lds((size_t)h->fw.plan.total_bytes + 64, (char)0xA5);   // garbage, like the device's LDS at kernel start
  ks::FastArgs a{h->pv, h->ws, h->fw};
  be_h2d(h, h->d_fast_args, &a, sizeof(a));
  const ks::FastArgs* a_ = h->d_fast_args;
  const int gs = h->fw.plan.global_state;
  if (h->fw.plan.helper) {
    // ksolve_pack_fast2: the placer's code; the refresher wavefront is a function the placer calls when it waits for it
    ks::FastHot* hs = (ks::FastHot*)(lds.data() + h->fw.plan.off_hot);
    ks::fast_mail_init(&hs->mail);
    ks::fast_emu_helper_k<1>() = ks::FastHelperK<1>();
    { const char* v = getenv("KSOLVE_EMU_REFRESHER_EAGER"); ks::fast_emu_helper_eager() = v && v[0] == '1'; }
    ks::FastEngine<ks::Wave, 0, 1, true> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve();
  } else if (h->fw.plan.rows == 1) {
    if (gs == 2) { ks::FastEngine<ks::Wave, 2, 1> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve(); }
    else if (gs) { ks::FastEngine<ks::Wave, 1, 1> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve(); }
    else { ks::FastEngine<ks::Wave, 0, 1> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve(); }
  } else {
    if (gs == 2) { ks::FastEngine<ks::Wave, 2, ks::kFastRows> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve(); }
    else if (gs) { ks::FastEngine<ks::Wave, 1, ks::kFastRows> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve(); }
    else { ks::FastEngine<ks::Wave, 0, ks::kFastRows> eng(&a_->pv, &a_->ws, &a_->fw, lds.data()); eng.solve(); }
  }
}
static void be_launch_pack_topo(ksolve_handle* h) {
  std::vector<char> lds((size_t)h->tw.plan.total_bytes + 64, (char)0xA5);   // garbage, like the device's LDS at kernel start
  ks::TopoArgs a{h->pv, h->ws, h->fw, h->tw};
  be_h2d(h, h->d_topo_args, &a, sizeof(a));
  const ks::TopoArgs* a_ = h->d_topo_args;
  ks::TopoEngine<ks::Wave> eng(&a_->pv, &a_->ws, &a_->fw, &a_->tw, lds.data());
  eng.solve();
}
static ks::FastQueueArgs fast_queue_args(ksolve_handle* h) {
  return ks::FastQueueArgs{h->pv.sorted_pods, h->pv.row_class, h->fw.q_class, h->fw.q_claim, h->fw.q_cnt, h->ws.assign, h->ws.slot, h->fw.cls_first, h->fw.cls_last, h->fw.max_active};
}
static void be_launch_fast_records(ksolve_handle* h, int n_claims) {
  ks::FastRecordArgs a{h->pv, h->ws, h->fw};
  for (int c = 0; c < n_claims; ++c) ks::fast_record_body<ks::Wave>(c, a);
  const ks::FastQueueArgs q = fast_queue_args(h);
  for (int i = 0; i < (int)h->n_pods; ++i) ks::fast_scatter_body(i, q);
}
static void be_launch_fast_queue(ksolve_handle* h, bool count_live) {
  const ks::FastQueueArgs q = fast_queue_args(h);
  for (int i = 0; i < (int)h->n_pods; ++i) ks::fast_queue_body(i, q);
  const int nc = (int)h->n_classes;
  if (count_live && h->fw.enabled && nc > 64 && nc <= 32768) for (int c = 0; c < nc; ++c) ks::fast_overlap_body(c, nc, q);
  for (int i = 0; i < (int)h->n_pods; ++i) ks::fast_mark_body(i, q);
}
static void be_launch_pack_fast_batch(ksolve_handle** hs, int n) {
  for (int i = 0; i < n; ++i) { be_tic(hs[i], ksi::T_PACK); be_launch_pack_fast(hs[i]); be_toc(hs[i], ksi::T_PACK); }
}
static void be_launch_pack_batch(ksolve_handle** hs, int n) {
  for (int i = 0; i < n; ++i) { be_tic(hs[i], ksi::T_PACK); be_launch_pack(hs[i]); be_toc(hs[i], ksi::T_PACK); }
}
static int be_device_available() { return 1; }
static int be_device_of(const ksolve_handle* h) { return (int)h->opts.device; }   // the emulation has as many "devices" as the options name
static void be_free(ksolve_handle* h, void* p) {
  auto it = std::find(h->allocations.begin(), h->allocations.end(), p);
  if (it != h->allocations.end()) h->allocations.erase(it);
  free(p);
}
static void be_launch_node_dead0(ksolve_handle*, int n_blocks, const ks::NodeDeadArgs& a) { for (int b = 0; b < n_blocks; ++b) for (int c = 0; c * ks::kDead0Classes < a.n_classes; ++c) ks::node_dead0_body<ks::Wave>(b, c, a); }
static void be_launch_claim_gather(ksolve_handle*, int n, const ks::ClaimGatherArgs& a) { for (int i = 0; i < n; ++i) ks::claim_gather_body(i, a); }
static void be_launch_sweep_items(ksolve_handle*, int n, const ks::SweepItemArgs& a) { for (int i = 0; i < n; ++i) ks::sweep_items_body(i, a); }
static void be_launch_pack_sweep(ksolve_handle* h, const ks::ProblemView* d_pv, ks::Workspace* d_items, int n, const ks::LdsPlan& plan_in, const uint32_t* d_order, uint32_t* d_next) {
  be_tic(h, ksi::T_PACK);
  if (plan_in.waves == 4) {
    // the compact form (ksolve_pack_sweep4): workgroups of four wavefronts over one LDS segment each — wave 0 prepares the shared
    // tables with the first probe it takes — and the probes handed out through the launch's counter in `d_order`. The emulated
    // wavefronts take one probe at a time in turn, so that each runs several probes on its own working set beside its neighbours'.
    typedef ks::Engine<ks::Wave, true, false, ks::ScratchSmall> Eng;
#if defined(__SANITIZE_ADDRESS__)
    // AddressSanitizer build: every wavefront's working set gets a poisoned tail, so that a working set outgrowing its slice of the
    // workgroup's LDS (into its neighbour's Scratch) is reported
    ks::LdsPlan plan = plan_in;
    plan.wave_stride += 64; plan.total_bytes += 4 * 64;
#else
    const ks::LdsPlan& plan = plan_in;
#endif
    const int grid = std::max(1, std::min((n + 3) / 4, 3));
    auto fetch = [&]() -> int { const uint32_t i = (*d_next)++; return i < (uint32_t)n ? (int)d_order[i] : -1; };
    std::vector<std::vector<char>> lds((size_t)grid);
    std::vector<uint32_t> active((size_t)grid, 0);
    for (int b = 0; b < grid; ++b) {
      lds[b].assign((size_t)plan.total_bytes + 64, (char)0xA5);   // garbage, like the device's LDS at kernel start
#if defined(__SANITIZE_ADDRESS__)
      for (int w = 0; w < 4; ++w) __asan_poison_memory_region(lds[b].data() + (plan.total_bytes - (4 - w) * plan.wave_stride) + plan.wave_stride - 64, 64);
#endif
      const int p = fetch();
      if (p < 0) continue;
      ks::LdsTables t0; t0.bind(lds[b].data(), plan, 0);
      { Eng eng(*d_pv, d_items[p], t0); active[b] = eng.prepare(); }
      Eng eng(*d_pv, d_items[p], t0);
      eng.solve(&active[b]);
    }
    for (bool more = true; more;) {
      more = false;
      for (int b = 0; b < grid; ++b) for (int wave = 0; wave < 4; ++wave) {
        const int p = fetch();
        if (p < 0) continue;
        more = true;
        ks::LdsTables tables; tables.bind(lds[b].data(), plan, (wave + 1) & 3);   // wave 0 had the first turn
        Eng eng(*d_pv, d_items[p], tables);
        eng.solve(&active[b]);
      }
    }
    be_toc(h, ksi::T_PACK);
    return;
  }
  const ks::LdsPlan& plan = plan_in;
  for (int p = 0; p < n; ++p) {
    std::vector<char> lds((size_t)plan.total_bytes + 64, (char)0xA5);   // garbage, like the device's LDS at kernel start
    ks::LdsTables tables;
    tables.bind(lds.data(), plan);
    ks::Engine<ks::Wave, true> eng(*d_pv, d_items[p], tables);
    eng.solve();
  }
  be_toc(h, ksi::T_PACK);
}

extern "C" {
ksolve_status ksolve_create(const ksolve_problem_desc* desc, const ksolve_options* opts, ksolve_handle** out) {
  ksolve_handle* h = new ksolve_handle();
  h->backend = new EmuBackend();
  ksolve_status s = ksi::create(desc, opts, h);
  *out = h;  // returned even on failure so the caller can read ksolve_last_error
  return s;
}
ksolve_status ksolve_probe_create(ksolve_handle* base, const ksolve_probe* probe, ksolve_handle** out) {
  ksolve_handle* h = new ksolve_handle();
  h->backend = new EmuBackend();
  *out = h;
  return ksi::probe_create(base, probe, h);
}
ksolve_status ksolve_solve(ksolve_handle* h, ksolve_results* out) { return ksi::solve(h, out); }
ksolve_status ksolve_sweep(ksolve_handle* base, const ksolve_sweep_desc* desc, ksolve_sweep_results* out) { return ksi::sweep(base, desc, out); }
ksolve_status ksolve_sweep_replicas(ksolve_handle** bases, uint32_t n_bases, const ksolve_sweep_desc* desc, ksolve_sweep_results* out) { return (bases && n_bases && out) ? ksi::sweep_replicas(bases, n_bases, desc, out) : KSOLVE_ERR_INVALID; }
void ksolve_sweep_results_free(ksolve_sweep_results* r) { if (r && r->impl) { delete (ksi::SweepImpl*)r->impl; r->impl = nullptr; } }
ksolve_status ksolve_solve_batch(ksolve_handle** hs, uint32_t n, ksolve_results* outs) { return ksi::solve_batch(hs, n, outs); }
ksolve_status ksolve_packing_vector(const ksolve_handle* h, const ksolve_results* r, double* count, double* cost) { return ksi::packing_vector(h, r->claims, count, cost); }
ksolve_status ksolve_packing_vector_sum(ksolve_handle* const* hs, const ksolve_results* rs, uint32_t n, double* count, double* cost) { return ksi::packing_vector_sum(hs, rs, n, count, cost); }
ksolve_status ksolve_cancel(ksolve_handle* h) { if (h->d_cancel) __atomic_store_n(h->d_cancel, 1, __ATOMIC_RELAXED); return KSOLVE_OK; }
void ksolve_results_free(ksolve_results* r) { if (r && r->impl) { delete (ksi::ResultsImpl*)r->impl; r->impl = nullptr; } }
void ksolve_destroy(ksolve_handle* h) {
  if (!h) return;
  for (void* p : h->allocations) free(p);
  delete (EmuBackend*)h->backend;
  delete h;
}
const char* ksolve_last_error(const ksolve_handle* h) { return h ? h->error.c_str() : "null handle"; }
uint32_t ksolve_abi_version(void) { return KSOLVE_ABI_VERSION; }
int ksolve_device_available(void) { return 0; }  // the emulation is not a device
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
int ksolve_is_emulation(void) { return 1; }
// The product's Go-sort (csrc/go_sort.h, used by the finalize kernel for OrderByPrice) on an array of integer keys:
// out_perm receives the original indices in sorted order, ties as Go's sort.Slice leaves them.
void ksolve_emu_go_sort(const long long* keys, int n, int* out_perm) {
  std::vector<long long> k(keys, keys + n);
  for (int i = 0; i < n; ++i) out_perm[i] = i;
  long long* kp = k.data();
  ks::go_sort_slice(n, [kp](int i, int j) { return kp[i] < kp[j]; },
                    [kp, out_perm](int i, int j) { std::swap(kp[i], kp[j]); std::swap(out_perm[i], out_perm[j]); });
}
// Drives the device's claim-order emulation (pdq_emul.h) with a trace of commits; out_ids receives the final order.
int ksolve_emu_order_trace(const int* ops, int n_ops, int* out_ids, unsigned long long* slow_sorts) {
  std::vector<uint32_t> key(n_ops + 1), ord(n_ops + 1), pos(n_ops + 1);
  ks::ClaimOrder<ks::Wave> o;
  o.key = key.data(); o.ord = ord.data(); o.pos = pos.data();
  int n_claims = 0;
  for (int i = 0; i < n_ops; ++i) {
    o.sort();
    if (ops[i] < 0) o.append(n_claims++);
    else o.increment(ops[i]);
  }
  o.sort();
  for (int i = 0; i < o.n; ++i) out_ids[i] = (int)ord[i];
  if (slow_sorts) *slow_sorts = o.slow_sorts;
  return o.n;
}
}
