// ksolve_impl.h — implementation of the C ABI (include/ksolve.h) on top of a tiny backend layer.
// Included by exactly one translation unit per build:
//   ksolve.hip      (product): HIP backend — device arena, one stream per handle, HIP events around every phase
//   tests/emu/ksolve_emu.cpp (TEST ONLY): host emulation of the same kernels, to fuzz the device algorithm against the
//                              oracle on machines without a GPU. Never built into, or loaded by, the product.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <array>
#include <thread>
#include <vector>

#include "../../include/ksolve.h"
#include "engine.h"
#include "kernels.h"
#include "fast_engine.h"
#include "topo_types.h"

// Backend contract (provided by the including TU):
//   void* be_alloc(ksolve_handle*, size_t bytes)  — zero-initialised device memory owned by the handle
//   void  be_h2d(ksolve_handle*, void* dst, const void* src, size_t), be_d2h(...), be_fill(ksolve_handle*, void*, int byte, size_t)
//   void  be_sync(ksolve_handle*)
//   launchers: be_launch_it_index, be_launch_row_hash, ..., be_launch_pack, be_sort_pods
//   timers: be_tic(h, slot), be_toc(h, slot) -> accumulate milliseconds per named phase
struct ksolve_handle;

namespace ksi {

using namespace ks;

struct HostReqTable {  // device pointers of an uploaded ksolve_reqsets
  ReqTable t{};
};

struct Timers {
  double ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
enum { T_UPLOAD = 0, T_INDEX = 1, T_CLASSIFY = 2, T_SORT = 3, T_PACK = 4, T_FINALIZE = 5, T_DOWNLOAD = 6, T_ROWHASH = 7 };   // T_ROWHASH: the streaming kernel of the classing phase alone (inside T_CLASSIFY)

}  // namespace ksi

struct ksolve_handle {
  std::string error;
  ksolve_options opts{};
  ksi::Timers timers;
  // problem sizes
  uint32_t n_keys = 0, req_words = 0, n_res = 0, n_its = 0, it_words = 0, n_templates = 0, n_pods = 0, n_rows = 0, n_classes = 0;
  uint32_t max_claims = 0, claim_words = 0, class_capacity = 0, n_nodes = 0;
  int n_kv = 0;
  ks::ProblemView pv{};
  ks::Workspace ws{};
  // device buffers needed by the host between phases
  ks::ItIndexArgs it_args{};
  ks::RowArgs row_args{};
  ks::SortKeyArgs sort_args{};
  uint32_t *d_idx_a = nullptr, *d_idx_b = nullptr;
  uint64_t *d_key_a = nullptr, *d_key_b = nullptr;
  double* d_cheapest = nullptr;
  int64_t* d_daemon_requests = nullptr;   // [max_claims][n_res] addDaemonRequests, by the finalize kernel
  // Results.TruncateInstanceTypes (only when ksolve_options.truncate_instance_types > 0), sized by the claims of the last solve
  int32_t* d_sort_idx = nullptr; double* d_sort_price = nullptr; uint32_t* d_ordered_count = nullptr; uint8_t* d_trunc_failed = nullptr;
  size_t trunc_capacity = 0;
  int* d_cancel = nullptr;
  ks::MutReqTable d_cls_reqs{}, d_cls_strict{};
  std::vector<void*> allocations;
  void* backend = nullptr;
  volatile int cancel_requested = 0;
  bool has_topology = false;
  ks::LdsPlan lds_big{};   // LDS plan of the BIG engine, used once a solve overflowed the LDS-resident claim order
  bool big_capable = false;
  int lite_saved = 0;
  uint32_t fast_live = 0;       // cursor engine: the most pod classes live at once in the queue (ksolve_fast_overlap)
  bool fast_live_known = false; uint32_t fast_live_classes = 0;   // ... counted by the handle's first solve (for that many classes)
  ks::FastWork fw{};            // cursor engine (fast_engine.h): workspace + LDS plan; fw.enabled while the problem may qualify
  uint32_t engine_used = 0, fast_reason = 0, fast_attempts = 0;
  ks::FastArgs* d_fast_args = nullptr;   // the record ksolve_pack_fast reads its problem from
  ks::TopoWork tw{};            // spread engine (topo_engine.h): workspace + LDS plan; tw.enabled while the problem may qualify (it borrows fw's buffers)
  ks::TopoArgs* d_topo_args = nullptr;
  // probes of a resident cluster (ksolve_probe_create): a probe handle shares the base's device tables
  ksolve_handle* base = nullptr;         // non-null: this handle is a probe of `base`
  bool prepared = false;                 // base: phases 1-3 have run and h_rank is valid
  std::vector<uint32_t> h_rank;          // base: queue position of every pod (queue.go:72-108 order)
  // a probe handle is its descriptor; ksolve_solve / ksolve_solve_batch run it through the sweep machinery (sweep_run)
  std::vector<uint64_t> h_it_off_avail, h_value_is_int;
  std::vector<double> h_it_off_price;
  std::vector<int64_t> h_value_int;
  int h_n_zones = 0, h_n_cts = 0;
  uint32_t pv_entries = 0;               // volume entries of all pods (capacity of Workspace::pv_log)
  std::vector<uint32_t> h_pod_pv_first;  // base: ProblemView::pod_pv_first on the host (a probe's log is sized by its own pods)
  std::vector<uint32_t> pr_nodes, pr_pods;
  std::vector<int64_t> pr_limits;
  // base: what every sweep shares — rejections of the pristine nodes per class, the consolidateAfter bitmap, the view in HBM,
  // and the arena the probes' workspaces are carved from (kept between sweeps, grown when a sweep needs more)
  bool resident = false;                 // ksolve_problem_desc.pod_node given: the problem is a whole cluster, solved through probes only
  bool sweep_ready = false;
  ks::ProblemView* d_pv = nullptr;
  char* sweep_arena = nullptr; size_t sweep_arena_bytes = 0;
  uint32_t fast_mc = 0;           // max_claims the cursor engine's plans are cut to
  double fast_need = 0;           // lower bound of the NodeClaims the batch needs (total requests / largest allocatable): picks the cursor engine's first plan
  double dead0_us = 0;            // ksolve_node_dead0 (every class x every pristine node), once per resident cluster
  bool sweep_arena_refused = false;   // the last sweep_run stopped because the device refused its arena (sweep() then halves the launch)
  size_t sweep_last_total = 0;   // arena bytes the last sweep_run laid out (held against sweep_probe_bytes by the test builds)
  char* sweep_fin = nullptr; size_t sweep_fin_bytes = 0;     // finalize outputs + gathered claim records of a sweep
};

// ---- backend hooks (defined by the including TU before this point is instantiated) ----
static void* be_alloc(ksolve_handle* h, size_t bytes);
static void* be_try_alloc(ksolve_handle* h, size_t bytes);   // the same, but a refusal is not an error of the handle: null, nothing recorded (the sweep arena: the caller retries smaller)
static void be_h2d(ksolve_handle* h, void* dst, const void* src, size_t bytes);
static void be_d2h(ksolve_handle* h, void* dst, const void* src, size_t bytes);
static void be_fill(ksolve_handle* h, void* dst, int byte, size_t bytes);
static void be_sync(ksolve_handle* h);
static void* be_stage(ksolve_handle* h, size_t bytes);   // page-locked host memory kept by the handle (valid until the next be_stage of the handle)
static bool be_ok(ksolve_handle* h);
static void be_tic(ksolve_handle* h, int slot);
static void be_toc(ksolve_handle* h, int slot);
static void be_range_drop(ksolve_handle* h);   // closes the trace range of a be_tic whose phase ends in an error (no timing)
// a phase that has several error exits: the range be_tic opened is closed on whichever way out (the marker trace nested wrongly
// for the rest of the process after a failed sweep)
struct PhaseRange { ksolve_handle* h; bool open; ~PhaseRange() { if (open) be_range_drop(h); } };
static void be_launch_it_index(ksolve_handle* h, int n, const ks::ItIndexArgs& a);
static void be_launch_row_hash(ksolve_handle* h, int n, const ks::RowArgs& a);
static void be_launch_row_class(ksolve_handle* h, int n, const ks::RowArgs& a);
static void be_launch_class_gather(ksolve_handle* h, int n, const ks::RowArgs& a);
static void be_sort_pods(ksolve_handle* h);  // fills ws/pv.sorted_pods
static void be_launch_pack(ksolve_handle* h);
static void be_launch_pack_fast(ksolve_handle* h);                 // one wavefront: FastEngine::solve
static void be_launch_pack_topo(ksolve_handle* h);                 // one wavefront: TopoEngine::solve
static void be_launch_pack_fast_batch(ksolve_handle** hs, int n);   // block b = the cursor engine on problem b; sets every handle's T_PACK timer
static void be_launch_fast_records(ksolve_handle* h, int n_claims); // one wavefront per claim: fast_record_body; then fast_scatter_body per queue entry
static void be_launch_fast_queue(ksolve_handle* h, bool count_live);   // one thread per queue entry: fast_queue_body (+ ksolve_fast_overlap when asked)
static void be_launch_pack_batch(ksolve_handle** hs, int n);
static void be_thread_init(ksolve_handle* h);   // makes the handle's device current on a worker thread   // one block per handle; sets every handle's T_PACK timer
static void be_launch_finalize(ksolve_handle* h, int n, const ks::FinalizeArgs& a);
static int be_device_available();
static int be_device_of(const ksolve_handle* h);     // the device ordinal the handle lives on
static void be_free(ksolve_handle* h, void* p);   // releases one be_alloc'ed block before the handle goes
static void be_launch_node_dead0(ksolve_handle* h, int n_blocks, const ks::NodeDeadArgs& a);   // one wavefront per 64 nodes
static void be_launch_pack_sweep(ksolve_handle* h, const ks::ProblemView* d_pv, ks::Workspace* d_items, int n, const ks::LdsPlan& plan, const uint32_t* d_order, uint32_t* d_next);   // block b = the general engine on probe b; sets T_PACK
static void be_launch_claim_gather(ksolve_handle* h, int n, const ks::ClaimGatherArgs& a);
static void be_launch_sweep_items(ksolve_handle* h, int n, const ks::SweepItemArgs& a);

namespace ksi {

template <class T>
static T* up(ksolve_handle* h, const T* src, size_t n) {
  if (n == 0) n = 1;
  T* d = (T*)be_alloc(h, n * sizeof(T));
  if (src) be_h2d(h, d, src, n * sizeof(T));
  return d;
}
template <class T>
static T* dz(ksolve_handle* h, size_t n) { return (T*)be_alloc(h, (n ? n : 1) * sizeof(T)); }

static ReqTable upload_reqs(ksolve_handle* h, const ksolve_reqsets& r, uint32_t n, uint32_t req_words, uint32_t n_keys, bool with_minv = true) {
  ReqTable t{};
  t.mask = up(h, r.mask, (size_t)n * req_words);
  t.defined = up(h, r.defined, n);
  t.complement = up(h, r.complement, n);
  t.has_gte = r.has_gte ? up(h, r.has_gte, n) : dz<uint32_t>(h, n);
  t.has_lte = r.has_lte ? up(h, r.has_lte, n) : dz<uint32_t>(h, n);
  t.gte = r.gte ? up(h, r.gte, (size_t)n * n_keys) : dz<int64_t>(h, (size_t)n * n_keys);
  t.lte = r.lte ? up(h, r.lte, (size_t)n * n_keys) : dz<int64_t>(h, (size_t)n * n_keys);
  if (!with_minv) t.minv = nullptr;
  else if (r.min_values) t.minv = up(h, r.min_values, (size_t)n * n_keys);
  else { int32_t* m = dz<int32_t>(h, (size_t)n * n_keys); be_fill(h, m, 0xFF, (size_t)n * n_keys * sizeof(int32_t)); t.minv = m; }
  return t;
}
static MutReqTable alloc_reqs(ksolve_handle* h, uint32_t n, uint32_t req_words, uint32_t n_keys) {
  MutReqTable t{};
  t.mask = dz<uint64_t>(h, (size_t)n * req_words);
  t.defined = dz<uint32_t>(h, n); t.complement = dz<uint32_t>(h, n); t.has_gte = dz<uint32_t>(h, n); t.has_lte = dz<uint32_t>(h, n);
  t.gte = dz<int64_t>(h, (size_t)n * n_keys); t.lte = dz<int64_t>(h, (size_t)n * n_keys); t.minv = dz<int32_t>(h, (size_t)n * n_keys);
  return t;
}
static ReqTable as_const(const MutReqTable& m) {
  ReqTable t{};
  t.mask = m.mask; t.defined = m.defined; t.complement = m.complement; t.has_gte = m.has_gte; t.has_lte = m.has_lte;
  t.gte = m.gte; t.lte = m.lte; t.minv = m.minv;
  return t;
}

static ksolve_status fail(ksolve_handle* h, ksolve_status s, const std::string& msg) {
  h->error = msg;
  return s;
}

static bool any_nonzero(const uint32_t* p, uint32_t n) {
  if (!p) return false;
  for (uint32_t i = 0; i < n; ++i) if (p[i]) return true;
  return false;
}

static void fast_plan_set(ksolve_handle* h, int plan, int rows);   // the cursor engine's memory plan (0 LDS / 1 claim state in HBM / 2 order arrays too)
static ksolve_status create(const ksolve_problem_desc* d, const ksolve_options* o, ksolve_handle* h) {
  if (!d || d->abi_version != KSOLVE_ABI_VERSION) return fail(h, KSOLVE_ERR_INVALID, "abi version mismatch");
  if (d->n_keys == 0 || d->n_keys > KSOLVE_MAX_KEYS) return fail(h, KSOLVE_ERR_INVALID, "n_keys out of range");
  if (d->n_res < 2 || d->n_res > KSOLVE_MAX_RES) return fail(h, KSOLVE_ERR_INVALID, "n_res must be in [2,8] (cpu, memory first)");
  if (d->n_templates > KSOLVE_MAX_TEMPLATES) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 32 NodeClaimTemplates");
  if (d->n_zones > KSOLVE_MAX_ZONES || d->n_captypes > KSOLVE_MAX_CAPTYPES) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 16 offering zones or 4 capacity types");
  if (d->n_its == 0 && d->n_templates) return fail(h, KSOLVE_ERR_INVALID, "templates without instance types");
  const uint32_t req_words = d->key_word_off[d->n_keys];
  if (req_words > (uint32_t)ks::kMaxReqWords) return fail(h, KSOLVE_ERR_UNSUPPORTED, "requirement dictionaries need more than 96 mask words");
  const uint32_t it_words = (d->n_its + 63) / 64;
  if (it_words > (uint32_t)ks::kMaxItWords) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 2048 instance types");
  if (d->key_instance_type < 0) return fail(h, KSOLVE_ERR_INVALID, "key_instance_type required");
  if (d->key_word_off[d->key_instance_type + 1] - d->key_word_off[d->key_instance_type] != it_words)
    return fail(h, KSOLVE_ERR_INVALID, "instance-type key dictionary must be the instance type list");
  if (d->n_nodes) {
    if (any_nonzero(d->node_reqs.has_gte, d->n_nodes) || any_nonzero(d->node_reqs.has_lte, d->n_nodes)) return fail(h, KSOLVE_ERR_INVALID, "existing-node requirements are label sets: no bounds");
  }
  if (d->topo.n > KSOLVE_MAX_TOPO_GROUPS) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 1024 topology groups");
  if (d->topo.n && (!d->pod_topo_owned || !d->pod_topo_selected)) return fail(h, KSOLVE_ERR_INVALID, "topology groups without pod_topo_owned / pod_topo_selected");
  for (uint32_t i = 0; i < d->n_its; ++i) {
    // reserved offerings would need the ReservationManager; capacity type index >= n_captypes never occurs by construction
    (void)i;
  }
  if (o) h->opts = *o;
  else { h->opts = ksolve_options{}; h->opts.max_steps = -1; }
  h->n_keys = d->n_keys; h->req_words = req_words; h->n_res = d->n_res; h->n_its = d->n_its; h->it_words = it_words;
  h->n_templates = d->n_templates; h->n_pods = d->n_pods; h->n_rows = d->n_pod_rows;
  if (h->n_rows < h->n_pods) return fail(h, KSOLVE_ERR_INVALID, "n_pod_rows < n_pods");

  be_tic(h, T_UPLOAD);
  ks::ProblemView& P = h->pv;
  ks::Dict& dict = P.dict;
  dict.n_keys = d->n_keys; dict.req_words = req_words;
  for (uint32_t k = 0; k <= d->n_keys; ++k) dict.key_word_off[k] = d->key_word_off[k];
  dict.well_known_mask = d->well_known_mask;
  dict.key_it = d->key_instance_type; dict.key_zone = d->key_zone; dict.key_ct = d->key_capacity_type; dict.key_hostname = d->key_hostname;
  dict.value_int = up(h, d->value_int, (size_t)req_words * 64);
  dict.value_is_int = up(h, d->value_is_int, req_words);
  // valid dictionary bits are not part of the ABI: a value is valid when some entity mentions it; complements
  // enumerate the dictionary, so derive validity from the union of every uploaded mask.
  {
    std::vector<uint64_t> valid(req_words, 0);
    auto acc = [&](const ksolve_reqsets& r, uint32_t n) { if (r.mask) for (size_t i = 0; i < (size_t)n * req_words; ++i) valid[i % req_words] |= r.mask[i]; };
    acc(d->it_reqs, d->n_its); acc(d->tmpl_reqs, d->n_templates); acc(d->pod_reqs, d->n_pod_rows); acc(d->pod_strict_reqs, d->n_pod_rows);
    acc(d->node_reqs, d->n_nodes);
    if (d->pod_volume_first) acc(d->volume_reqs, d->n_volume_reqs);
    if (d->topo.n) {
      acc(d->topo.filter_reqs, d->topo.filter_first[d->topo.n]);
      for (uint32_t g = 0; g < d->topo.n; ++g) if (d->topo.key[g] >= 0) {
        const uint32_t w0 = d->key_word_off[d->topo.key[g]], nw = d->key_word_off[d->topo.key[g] + 1] - w0;
        for (uint32_t x = 0; x < nw && x < d->topo.domain_words; ++x) valid[w0 + x] |= d->topo.domains[(size_t)g * d->topo.domain_words + x];
      }
    }
    // every instance type name is a valid value of the instance-type key
    for (uint32_t i = 0; i < d->n_its; ++i) valid[d->key_word_off[d->key_instance_type] + i / 64] |= 1ull << (i % 64);
    dict.value_valid = up(h, valid.data(), req_words);
  }
  P.n_res = d->n_res; P.n_its = d->n_its; P.it_words = it_words;
  P.it_alloc = up(h, d->it_allocatable, (size_t)d->n_res * d->n_its);
  P.it_cap = up(h, d->it_capacity, (size_t)d->n_res * d->n_its);
  P.it_off_avail = up(h, d->it_offering_avail, d->n_its);
  P.it_off_price = up(h, d->it_offering_price, (size_t)d->n_its * 64);
  // host copies for ksolve_packing_vector (the per-instance-type summary of a Results is computed where the Results are)
  h->h_it_off_avail.assign(d->it_offering_avail, d->it_offering_avail + d->n_its);
  h->h_it_off_price.assign(d->it_offering_price, d->it_offering_price + (size_t)d->n_its * 64);
  h->h_value_int.assign(d->value_int, d->value_int + (size_t)req_words * 64);
  h->h_value_is_int.assign(d->value_is_int, d->value_is_int + req_words);
  h->h_n_zones = (int)d->n_zones; h->h_n_cts = (int)d->n_captypes;
  P.n_zones = d->n_zones; P.n_cts = d->n_captypes;
  P.n_xg = (int)d->n_override_groups; P.xg_it = nullptr; P.xg_alloc = nullptr; P.xg_avail = nullptr; P.it_base_avail = P.it_off_avail;
  for (int r = 0; r < 8; ++r) P.xg_bonus[r] = 0;
  if (d->n_override_groups) {
    // offering override groups (types.go:202-269)
    const uint32_t nx = d->n_override_groups;
    if (nx > KSOLVE_MAX_OVERRIDE_GROUPS) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than KSOLVE_MAX_OVERRIDE_GROUPS offering override groups");
    if (!d->override_it || !d->override_allocatable || !d->override_avail || !d->it_base_avail) return fail(h, KSOLVE_ERR_INVALID, "override groups without their arrays");
    for (uint32_t e = 0; e < nx; ++e) {
      if (d->override_it[e] >= d->n_its) return fail(h, KSOLVE_ERR_INVALID, "override group of an unknown instance type");
      if (d->override_avail[e] & ~d->it_offering_avail[d->override_it[e]]) return fail(h, KSOLVE_ERR_INVALID, "override group offering outside it_offering_avail");
      for (uint32_t r = 0; r < d->n_res; ++r) {
        const int64_t a = d->override_allocatable[(size_t)r * nx + e], b = d->it_allocatable[(size_t)r * d->n_its + d->override_it[e]];
        if (a > b && a - b > P.xg_bonus[r]) P.xg_bonus[r] = a - b;
      }
    }
    for (uint32_t i = 0; i < d->n_its; ++i) if (d->it_base_avail[i] & ~d->it_offering_avail[i]) return fail(h, KSOLVE_ERR_INVALID, "it_base_avail outside it_offering_avail");
    P.xg_it = up(h, d->override_it, nx);
    P.xg_alloc = up(h, d->override_allocatable, (size_t)d->n_res * nx);
    P.xg_avail = up(h, d->override_avail, nx);
    P.it_base_avail = up(h, d->it_base_avail, d->n_its);
  }
  P.it_reqs = upload_reqs(h, d->it_reqs, d->n_its, req_words, d->n_keys);
  uint64_t* kv_has = dz<uint64_t>(h, (size_t)req_words * 64 * it_words);
  uint64_t* key_undef = dz<uint64_t>(h, (size_t)d->n_keys * it_words);
  uint64_t* key_compl = dz<uint64_t>(h, (size_t)d->n_keys * it_words);
  uint64_t* key_neg = dz<uint64_t>(h, (size_t)d->n_keys * it_words);
  uint64_t* alloc_ok = dz<uint64_t>(h, it_words);
  P.kv_has = kv_has; P.key_undef = key_undef; P.key_compl = key_compl; P.key_neg = key_neg; P.it_alloc_ok = alloc_ok;
  h->it_args = ks::ItIndexArgs{dict, (int)d->n_its, (int)it_words, (int)d->n_res, P.it_reqs, P.it_alloc, kv_has, key_undef, key_compl, key_neg, alloc_ok, dz<uint32_t>(h, 1)};

  {
    // keys some instance type defines, and a compact row index for every dictionary value of those keys
    uint32_t it_keys = 0;
    for (uint32_t i = 0; i < d->n_its; ++i) it_keys |= d->it_reqs.defined[i];
    if (d->key_instance_type >= 0) it_keys &= ~(1u << d->key_instance_type);
    P.it_keys = it_keys;
    std::vector<uint16_t> slot((size_t)req_words * 64, 0xFFFF);
    std::vector<uint64_t> used(req_words, 0);
    for (uint32_t i = 0; i < d->n_its; ++i) for (uint32_t x = 0; x < req_words; ++x) used[x] |= d->it_reqs.mask[(size_t)i * req_words + x];
    int n_kv = 0;
    for (uint32_t k = 0; k < d->n_keys; ++k) {
      if (!((it_keys >> k) & 1)) continue;
      // complements on the instance-type side "have" every dictionary value they do not exclude: give every valid value a row
      bool any_compl = false;
      for (uint32_t i = 0; i < d->n_its; ++i) if ((d->it_reqs.complement[i] >> k) & 1) any_compl = true;
      for (uint32_t x = d->key_word_off[k]; x < d->key_word_off[k + 1]; ++x)
        for (int b = 0; b < 64; ++b) {
          bool want = (used[x] >> b) & 1;
          if (any_compl) want = true;
          if (want) { if (n_kv >= 0xFFFE) return fail(h, KSOLVE_ERR_UNSUPPORTED, "too many instance-type label values"); slot[(size_t)x * 64 + b] = (uint16_t)n_kv++; }
        }
    }
    P.kv_slot = up(h, slot.data(), slot.size());
    h->n_kv = n_kv;
  }

  P.n_templates = d->n_templates;
  P.tmpl_reqs = upload_reqs(h, d->tmpl_reqs, d->n_templates, req_words, d->n_keys);
  P.tmpl_taints = up(h, d->tmpl_taints, d->n_templates);
  P.tmpl_its = up(h, d->tmpl_its, (size_t)d->n_templates * it_words);
  P.tmpl_limit_mask = up(h, d->tmpl_limit_mask, d->n_templates);
  P.tmpl_limits = up(h, d->tmpl_limits, (size_t)d->n_templates * (d->n_res + 1));

  {
    // daemon-overhead groups; without daemonsets every template is one group with no overhead
    std::vector<int> first(d->n_templates + 1, 0);
    std::vector<int64_t> ov;
    std::vector<uint64_t> gits;
    uint64_t nonzero = 0, nonempty = 0;
    if (d->tmpl_daemon_first) {
      const uint32_t ng = d->tmpl_daemon_first[d->n_templates];
      if (ng > 64) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 64 daemon-overhead groups");
      for (uint32_t t = 0; t <= d->n_templates; ++t) first[t] = (int)d->tmpl_daemon_first[t];
      for (uint32_t t = 0; t < d->n_templates; ++t) if (first[t + 1] <= first[t]) return fail(h, KSOLVE_ERR_INVALID, "every template needs at least one daemon-overhead group");
      ov.assign(d->daemon_group_overhead, d->daemon_group_overhead + (size_t)ng * d->n_res);
      gits.assign(d->daemon_group_its, d->daemon_group_its + (size_t)ng * it_words);
      for (uint32_t g = 0; g < ng; ++g) {
        for (uint32_t r = 0; r < d->n_res; ++r) { if (ov[(size_t)g * d->n_res + r] < 0) return fail(h, KSOLVE_ERR_INVALID, "negative daemon overhead"); if (ov[(size_t)g * d->n_res + r]) nonzero |= 1ull << g; }
        if (d->daemon_group_nonempty && d->daemon_group_nonempty[g]) nonempty |= 1ull << g;
      }
    } else {
      for (uint32_t t = 0; t <= d->n_templates; ++t) first[t] = (int)t;
      ov.assign((size_t)std::max(1u, d->n_templates) * d->n_res, 0);
      gits.assign(d->tmpl_its, d->tmpl_its + (size_t)d->n_templates * it_words);
      if (gits.empty()) gits.assign(it_words, 0);
    }
    P.n_dg = first[d->n_templates];
    P.dg_first = up(h, first.data(), first.size());
    P.dg_ov = up(h, ov.data(), ov.size());
    P.dg_its = up(h, gits.data(), gits.size());
    P.dg_nonzero = nonzero; P.dg_nonempty = nonempty;
    // host ports: only matter when some pod binds one
    P.node_removed = nullptr;
  P.hp_on = d->pod_host_ports ? 1 : 0; P.cls_hp = nullptr; P.dg_hp = nullptr; P.node_hp0 = nullptr;
    if (P.hp_on) {
      if (!d->pod_host_port_conflicts) return fail(h, KSOLVE_ERR_INVALID, "pod_host_ports without pod_host_port_conflicts");
      std::vector<uint64_t> ghp((size_t)std::max(1, P.n_dg), 0);
      if (d->tmpl_daemon_first && d->daemon_group_host_ports) for (int g = 0; g < P.n_dg; ++g) ghp[g] = d->daemon_group_host_ports[g];
      P.dg_hp = up(h, ghp.data(), ghp.size());
      if (d->n_nodes && d->node_host_ports) P.node_hp0 = up(h, d->node_host_ports, d->n_nodes);
    }
  }

  {
    // reserved offerings
    P.n_resv = (int)d->n_reservations; P.key_rid = d->key_reservation_id; P.ct_reserved = d->captype_reserved;
    P.reserved_strict = h->opts.reserved_offering_strict ? 1 : 0;
    P.reserved_on = (h->opts.reserved_capacity && d->n_reservations > 0) ? 1 : 0;
    if (d->n_reservations > 64) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 64 capacity reservations");
    if (P.reserved_on) {
      if (!d->it_reserved_first || !d->reservation_capacity || d->key_reservation_id < 0 || d->captype_reserved < 0) return fail(h, KSOLVE_ERR_INVALID, "reserved offerings need it_reserved_first, reservation_capacity, key_reservation_id and captype_reserved");
      const uint32_t no = d->it_reserved_first[d->n_its];
      P.resv_cap0 = up(h, d->reservation_capacity, d->n_reservations);
      P.it_resv_first = up(h, d->it_reserved_first, (size_t)d->n_its + 1);
      P.resv_zone = up(h, d->reserved_zone, no); P.resv_id = up(h, d->reserved_id, no); P.resv_price = up(h, d->reserved_price, no);
    }
  }

  P.n_pods = d->n_pods; P.n_rows = d->n_pod_rows;
  P.row_next = up(h, d->pod_next_variant, d->n_pod_rows);
  P.pod_is_pending = up(h, d->pod_is_pending, d->n_pods);
  ks::RowArgs& R = h->row_args;
  R.dict = dict; R.n_rows = d->n_pod_rows; R.n_res = d->n_res;
  R.requests = up(h, d->pod_requests, (size_t)d->n_res * d->n_pod_rows);
  // Pod rows carry no minValues when they come from the reference's PodData (minValues belong to NodePool requirements): a
  // table of nils is neither uploaded nor streamed by the classing kernel (absent == nil everywhere it is read).
  auto all_nil = [&](const ksolve_reqsets& r) {
    if (!r.min_values) return true;
    const size_t n = (size_t)d->n_pod_rows * d->n_keys;
    for (size_t i = 0; i < n; ++i) if (r.min_values[i] >= 0) return false;
    return true;
  };
  const bool strict_same = d->pod_strict_reqs.mask == d->pod_reqs.mask || d->pod_strict_reqs.mask == nullptr;
  const bool rows_nil = all_nil(d->pod_reqs) && (strict_same || all_nil(d->pod_strict_reqs));
  R.reqs = upload_reqs(h, d->pod_reqs, d->n_pod_rows, req_words, d->n_keys, !rows_nil);
  P.strict_same = strict_same ? 1 : 0;
  if (strict_same) R.strict = R.reqs;
  else R.strict = upload_reqs(h, d->pod_strict_reqs, d->n_pod_rows, req_words, d->n_keys, !rows_nil);
  R.tolerates = up(h, d->pod_tolerates, d->n_pod_rows);
  // volume requirement alternatives (nodeclaim.go:138-157, existingnode.go:108-139)
  P.vol_on = d->pod_volume_first ? 1 : 0; P.cls_vol = nullptr; P.vol_reqs = ReqTable{};
  R.vol = nullptr; R.cls_vol = nullptr;
  if (P.vol_on) {
    if (!d->pod_volume_count || (d->n_volume_reqs && !d->volume_reqs.mask)) return fail(h, KSOLVE_ERR_INVALID, "pod_volume_first without pod_volume_count / volume_reqs");
    std::vector<uint64_t> vol(d->n_pod_rows);
    for (uint32_t r = 0; r < d->n_pod_rows; ++r) {
      const uint32_t f = d->pod_volume_first[r], c = d->pod_volume_count[r];
      if ((uint64_t)f + c > d->n_volume_reqs) return fail(h, KSOLVE_ERR_INVALID, "pod volume alternatives outside volume_reqs");
      vol[r] = c ? ((uint64_t)f | ((uint64_t)c << 32)) : 0ull;
    }
    if (d->volume_reqs.min_values) for (size_t i = 0; i < (size_t)d->n_volume_reqs * d->n_keys; ++i) if (d->volume_reqs.min_values[i] >= 0) return fail(h, KSOLVE_ERR_INVALID, "volume requirements carry no minValues");
    R.vol = up(h, vol.data(), vol.size());
    P.vol_reqs = upload_reqs(h, d->volume_reqs, d->n_volume_reqs ? d->n_volume_reqs : 1, req_words, d->n_keys);
  }
  R.host_ports = nullptr; R.cls_host_ports = nullptr;
  if (P.hp_on) {
    std::vector<uint64_t> hp((size_t)d->n_pod_rows * 2);
    for (uint32_t r = 0; r < d->n_pod_rows; ++r) { hp[(size_t)r * 2] = d->pod_host_ports[r]; hp[(size_t)r * 2 + 1] = d->pod_host_port_conflicts[r]; }
    R.host_ports = up(h, hp.data(), hp.size());
  }
  R.topo_words = (int)((d->topo.n + 63) / 64);
  R.topo_owned = d->topo.n ? up(h, d->pod_topo_owned, (size_t)d->n_pod_rows * R.topo_words) : nullptr;
  R.topo_selected = d->topo.n ? up(h, d->pod_topo_selected, (size_t)d->n_pod_rows * R.topo_words) : nullptr;
  h->has_topology = d->topo.n != 0;
  uint32_t ts = 64;
  while (ts < 2 * d->n_pod_rows) ts <<= 1;
  R.table_size = ts; R.seed = 0x6b73703176310a01ull;
  R.hash_keep = ~0ull;
#ifdef KSOLVE_TEST_HOOKS   // only the test builds of the library (tests/emu/) read test switches; the product binary has none
  if (const char* keep = getenv("KSOLVE_TEST_HASH_KEEP")) R.hash_keep = strtoull(keep, nullptr, 0);   // collision-detection test: distinct rows forced onto one hash must be reported, never merged
#endif
  R.table_hash = dz<uint64_t>(h, ts); R.table_rep = dz<uint32_t>(h, ts); R.table_class = dz<uint32_t>(h, ts);
  R.row_slot = dz<uint32_t>(h, d->n_pod_rows);
  uint32_t* row_class = dz<uint32_t>(h, d->n_pod_rows);
  R.row_class = row_class; P.row_class = row_class;
  R.n_classes = dz<uint32_t>(h, 1); R.collision = dz<uint32_t>(h, 1);
  int64_t* minreq = dz<int64_t>(h, d->n_res);
  R.min_request = minreq; P.min_request = minreq;

  // existing nodes: SoA (word-major masks, dimension-major remaining) so that 64 lanes probe 64 nodes coalesced
  {
    const uint32_t ne = d->n_nodes;
    P.n_nodes = (int)ne; P.node_words = (int)((ne + 63) / 64);
    h->n_nodes = ne;
    if (ne) {
      std::vector<uint64_t> tm((size_t)req_words * ne);
      for (uint32_t e = 0; e < ne; ++e) for (uint32_t x = 0; x < req_words; ++x) tm[(size_t)x * ne + e] = d->node_reqs.mask[(size_t)e * req_words + x];
      h->ws.n_mask0 = up(h, tm.data(), tm.size());
      h->ws.n_defined0 = up(h, d->node_reqs.defined, ne);
      h->ws.n_complement0 = up(h, d->node_reqs.complement, ne);
      h->ws.n_remaining0 = up(h, d->node_remaining, (size_t)d->n_res * ne);
      h->ws.n_mask = dz<uint64_t>(h, tm.size());
      h->ws.n_defined = dz<uint32_t>(h, ne); h->ws.n_complement = dz<uint32_t>(h, ne);
      if (any_nonzero(d->pod_reqs.has_gte, d->n_pod_rows) || any_nonzero(d->pod_reqs.has_lte, d->n_pod_rows)) {
        h->ws.n_hg = dz<uint32_t>(h, ne); h->ws.n_hl = dz<uint32_t>(h, ne);
        h->ws.n_gte = dz<int64_t>(h, (size_t)d->n_keys * ne); h->ws.n_lte = dz<int64_t>(h, (size_t)d->n_keys * ne);
      }
      h->ws.n_remaining = dz<int64_t>(h, (size_t)d->n_res * ne);
      h->ws.n_npods = dz<uint32_t>(h, ne);
      h->ws.n_hp = P.hp_on ? dz<uint64_t>(h, ne) : nullptr;
      P.node_taints = up(h, d->node_taints, ne);
      std::vector<uint8_t> fl(ne);
      for (uint32_t e = 0; e < ne; ++e) fl[e] = (uint8_t)((d->node_initialized && d->node_initialized[e] ? 1 : 0) | (d->node_under_consolidate_after && d->node_under_consolidate_after[e] ? 2 : 0));
      P.node_flags = up(h, fl.data(), ne);
    }
    P.pod_from_deleting = d->pod_from_deleting_node ? up(h, d->pod_from_deleting_node, d->n_pods) : nullptr;
    P.pod_node = d->pod_node ? up(h, d->pod_node, d->n_pods) : nullptr;
    h->resident = d->pod_node != nullptr;
    if (d->pod_node) for (uint32_t p = 0; p < d->n_pods; ++p) if (d->pod_node[p] >= (int32_t)ne) return fail(h, KSOLVE_ERR_INVALID, "pod_node out of range");
    // CSI volume limits of existing nodes (VolumeUsage)
    P.pv_on = 0; P.n_pv_drivers = 0;
    if (d->n_volume_drivers && ne) {
      if (d->n_volume_drivers > KSOLVE_MAX_VOLUME_DRIVERS) return fail(h, KSOLVE_ERR_UNSUPPORTED, "more than 8 CSI drivers with volume limits");
      if (!d->volume_driver || !d->pod_pv_first || !d->pod_pvs || !d->node_pv_first || !d->node_pvs || !d->node_pv_limit) return fail(h, KSOLVE_ERR_INVALID, "volume limit tables missing");
      for (uint32_t p = 0; p < d->n_pods; ++p) {
        if (d->pod_pv_first[p + 1] < d->pod_pv_first[p] || d->pod_pv_first[p + 1] - d->pod_pv_first[p] > 64) return fail(h, KSOLVE_ERR_UNSUPPORTED, "a pod with more than 64 volumes under CSI limits");
        for (uint32_t i = d->pod_pv_first[p]; i < d->pod_pv_first[p + 1]; ++i) if (d->pod_pvs[i] >= d->n_volumes) return fail(h, KSOLVE_ERR_INVALID, "pod volume id out of range");
      }
      for (uint32_t e = 0; e < ne; ++e) for (uint32_t i = d->node_pv_first[e]; i < d->node_pv_first[e + 1]; ++i) {
        if (d->node_pvs[i] >= d->n_volumes) return fail(h, KSOLVE_ERR_INVALID, "node volume id out of range");
        if (i > d->node_pv_first[e] && d->node_pvs[i] <= d->node_pvs[i - 1]) return fail(h, KSOLVE_ERR_INVALID, "node volume ids must ascend");
      }
      for (uint32_t v = 0; v < d->n_volumes; ++v) if (d->volume_driver[v] >= d->n_volume_drivers) return fail(h, KSOLVE_ERR_INVALID, "volume driver out of range");
      P.pv_on = 1; P.n_pv_drivers = (int)d->n_volume_drivers;
      P.pv_driver = up(h, d->volume_driver, d->n_volumes);
      P.pod_pv_first = up(h, d->pod_pv_first, (size_t)d->n_pods + 1);
      P.pod_pvs = up(h, d->pod_pvs, std::max<size_t>(1, d->pod_pv_first[d->n_pods]));
      P.node_pv_first = up(h, d->node_pv_first, (size_t)ne + 1);
      P.node_pvs = up(h, d->node_pvs, std::max<size_t>(1, d->node_pv_first[ne]));
      P.node_pv_limit = up(h, d->node_pv_limit, (size_t)ne * d->n_volume_drivers);
      h->ws.pv_log = dz<uint64_t>(h, std::max<size_t>(1, d->pod_pv_first[d->n_pods]));
      h->pv_entries = d->pod_pv_first[d->n_pods];
      h->h_pod_pv_first.assign(d->pod_pv_first, d->pod_pv_first + d->n_pods + 1);
    }
  }

  ks::SortKeyArgs& S = h->sort_args;
  S.n_pods = d->n_pods; S.n_rows = d->n_pod_rows; S.requests = R.requests;
  S.creation = up(h, d->pod_creation, d->n_pods);
  S.uid_hi = up(h, d->pod_uid_hi, d->n_pods); S.uid_lo = up(h, d->pod_uid_lo, d->n_pods);
  h->d_idx_a = dz<uint32_t>(h, d->n_pods); h->d_idx_b = dz<uint32_t>(h, d->n_pods);
  h->d_key_a = dz<uint64_t>(h, d->n_pods); h->d_key_b = dz<uint64_t>(h, d->n_pods);

  // workspace
  uint32_t mc = h->opts.max_claims ? h->opts.max_claims : d->n_pods;
  if (mc > d->n_pods) mc = d->n_pods;
  if (mc == 0) mc = 1;
  h->max_claims = mc; h->claim_words = (mc + 63) / 64;
  ks::Workspace& W = h->ws;
  W.max_claims = (int)mc; W.claim_words = (int)h->claim_words;
  ks::RecLayout lay{(int)req_words, (int)it_words, (int)d->n_res, (int)d->n_keys};
  P.lay = lay;
  W.c_hot = dz<uint64_t>(h, (size_t)mc * lay.c_hot_words());
  W.c_cold = dz<uint64_t>(h, (size_t)mc * lay.cold_words());
  // + 64: the headroom prefilter reads one lane per claim of a 64-claim word; when max_claims is not a multiple of 64
  // the lanes past the last claim of the last dimension's row read (and discard) up to 63 entries beyond it
  W.c_headroom = dz<int64_t>(h, (size_t)mc * d->n_res + 64);
  W.c_reserved = dz<uint64_t>(h, mc);
  W.c_hp = P.hp_on ? dz<uint64_t>(h, mc) : nullptr;
  W.o_key = dz<uint32_t>(h, mc); W.o_ord = dz<uint32_t>(h, mc); W.o_pos = dz<uint32_t>(h, mc);
  W.queue = dz<uint32_t>(h, (size_t)d->n_pods + 1); W.last_len = dz<uint32_t>(h, d->n_pods);
  W.t_its = dz<uint64_t>(h, (size_t)d->n_templates * it_words);
  W.t_remaining = dz<int64_t>(h, (size_t)d->n_templates * (d->n_res + 1));
  W.assign = dz<int32_t>(h, d->n_pods); W.err = dz<uint8_t>(h, d->n_pods); W.diag = dz<uint8_t>(h, d->n_pods); W.slot = dz<uint32_t>(h, d->n_pods);
  W.n_claims_out = dz<int>(h, 1); W.status_out = dz<int>(h, 1);
  h->d_cancel = dz<int>(h, 1);
  W.cancel_flag = h->d_cancel;
  W.max_steps = h->opts.max_steps;
  W.min_values_best_effort = h->opts.min_values_best_effort ? 1 : 0;
  W.counters = dz<ks::Counters>(h, 1);
  h->d_cheapest = dz<double>(h, mc);
  h->d_daemon_requests = dz<int64_t>(h, (size_t)mc * d->n_res);
  {
    // topology groups
    ks::TopoView& T = P.topo;
    const ksolve_topology& t = d->topo;
    const uint32_t G = t.n;
    T.n_groups = (int)G; T.dom_words = (int)std::max(1u, t.domain_words);
    T.words = (int)((G + 63) / 64);
    for (int w = 0; w < ks::kMaxTopoWords; ++w) { T.inverse_mask[w] = 0; T.initially_active[w] = 0; T.alias_mask[w] = 0; }
    T.n_host_groups = 0; T.n_alias = 0; T.alias_class = nullptr;
    if (G) {
      std::vector<int16_t> host_slot(G, -1);
      std::vector<int32_t> nonzero(G, 0);
      const size_t dv = (size_t)T.dom_words * 64;
      for (uint32_t g = 0; g < G; ++g) {
        if (t.type[g] > 2) return fail(h, KSOLVE_ERR_INVALID, "topology group type out of range");
        if (t.key[g] >= (int32_t)d->n_keys) return fail(h, KSOLVE_ERR_INVALID, "topology group key out of range");
        if (t.key[g] >= 0 && d->key_word_off[t.key[g] + 1] - d->key_word_off[t.key[g]] > t.domain_words) return fail(h, KSOLVE_ERR_INVALID, "topology domain_words too small for a group key");
        if (t.inverse[g]) T.inverse_mask[g >> 6] |= 1ull << (g & 63);
        if (t.initially_active[g] || t.inverse[g]) T.initially_active[g >> 6] |= 1ull << (g & 63);
        if (t.key[g] < 0) {
          host_slot[g] = (int16_t)T.n_host_groups++;
          if (t.init_node_counts) for (uint32_t e = 0; e < d->n_nodes; ++e) if (t.init_node_counts[(size_t)g * d->n_nodes + e] > 0) nonzero[g]++;
        } else if (t.init_counts) {
          for (size_t v = 0; v < dv; ++v) if (t.init_counts[(size_t)g * dv + v] > 0) nonzero[g]++;
        }
      }
      // topology keys whose dictionary fits one mask word get a per-claim "values still admitted" table for the scan prefilter
      std::vector<int8_t> key_slot(G, -1);
      T.n_key_slots = 0;
      for (uint32_t g = 0; g < G; ++g) {
        if (t.key[g] < 0 || d->key_word_off[t.key[g] + 1] - d->key_word_off[t.key[g]] != 1) continue;
        int sl = -1;
        for (int i = 0; i < T.n_key_slots; ++i) if (T.slot_key[i] == t.key[g]) sl = i;
        if (sl < 0 && T.n_key_slots < 4) { sl = T.n_key_slots++; T.slot_key[sl] = t.key[g]; }
        key_slot[g] = (int8_t)sl;
      }
      T.key_slot = up(h, key_slot.data(), G);
      W.c_keymask = dz<uint64_t>(h, (size_t)std::max(1, T.n_key_slots) * mc);
      W.kv_claims = dz<uint64_t>(h, (size_t)std::max(1, T.n_key_slots) * 64 * h->claim_words);
      T.type = up(h, t.type, G); T.key = up(h, t.key, G); T.host_slot = up(h, host_slot.data(), G);
      T.max_skew = up(h, t.max_skew, G); T.min_domains = up(h, t.min_domains, G);
      T.domains0 = up(h, t.domains, (size_t)G * T.dom_words);
      T.counts0 = t.init_counts ? up(h, t.init_counts, (size_t)G * dv) : dz<int32_t>(h, (size_t)G * dv);
      std::vector<int32_t> nc((size_t)std::max(1, T.n_host_groups) * std::max(1u, d->n_nodes), 0);
      if (t.init_node_counts) for (uint32_t g = 0; g < G; ++g) if (host_slot[g] >= 0)
        for (uint32_t e = 0; e < d->n_nodes; ++e) nc[(size_t)host_slot[g] * d->n_nodes + e] = t.init_node_counts[(size_t)g * d->n_nodes + e];
      T.node_counts0 = up(h, nc.data(), nc.size());
      T.nonzero0 = up(h, nonzero.data(), G);
      T.f_affinity = up(h, t.filter_affinity_honor, G); T.f_taint = up(h, t.filter_taint_honor, G);
      T.f_first = up(h, t.filter_first, G + 1);
      T.f_reqs = upload_reqs(h, t.filter_reqs, t.filter_first[G], req_words, d->n_keys);
      T.f_tolerates = up(h, t.filter_tolerates, G);
      T.value_rank = up(h, t.value_rank, (size_t)req_words * 64);
      T.node_host_value = d->n_nodes ? (t.node_hostname_value ? up(h, t.node_hostname_value, d->n_nodes) : nullptr) : nullptr;
      if (d->n_nodes && !t.node_hostname_value) return fail(h, KSOLVE_ERR_INVALID, "topology with existing nodes needs node_hostname_value");
      T.dom_universe = t.domain_universe ? up(h, t.domain_universe, (size_t)G * T.dom_words) : nullptr;
      T.dom_regs0 = t.domain_node_regs ? up(h, t.domain_node_regs, (size_t)G * dv) : nullptr;
      W.tg_domains = dz<uint64_t>(h, (size_t)G * T.dom_words);
      W.tg_counts = dz<int32_t>(h, (size_t)G * dv);
      W.tg_node_counts = dz<int32_t>(h, nc.size());
      W.tg_claim_counts = dz<int32_t>(h, (size_t)std::max(1, T.n_host_groups) * mc);
      W.host_le = dz<uint64_t>(h, (size_t)std::max(1, T.n_host_groups) * 2 * h->claim_words);
      W.tg_nonzero = dz<int32_t>(h, G);
      if (t.alias_class && t.n_alias_classes) {
        std::vector<int16_t> ac(G, -1);
        for (uint32_t g = 0; g < G; ++g) {
          const int32_t c = t.alias_class[g];
          if (c < 0) continue;
          if ((uint32_t)c >= t.n_alias_classes || t.n_alias_classes > 32767) return fail(h, KSOLVE_ERR_INVALID, "topology alias class out of range");
          if (t.inverse[g] || t.initially_active[g]) return fail(h, KSOLVE_ERR_INVALID, "only groups created by relaxation can share a hash with different contents");
          ac[g] = (int16_t)c; T.alias_mask[g >> 6] |= 1ull << (g & 63);
        }
        T.n_alias = (int)t.n_alias_classes;
        T.alias_class = up(h, ac.data(), G);
        W.tg_alias_active = dz<int32_t>(h, t.n_alias_classes);
      }
    }
  }
  {
    // LDS plan of the pack kernel: tables first, the claim order gets what is left of the 160 KiB
    ks::LdsPlan& lp = P.lds;
    auto align = [](int x) { return (x + 15) & ~15; };
    const int np = (int)it_words * 64;
    int off = 0;
    lp.waves = 0; lp.wave_stride = 0; lp.off_shared_misc = 0;   // a one-wavefront plan (the compact sweep lays out its own, sweep_run)
    lp.off_alloc = off; off = align(off + (int)d->n_res * np * 8);
    lp.off_avail = off; off = align(off + np * 8);
    lp.n_kv = h->n_kv;
    lp.off_kv = off; off = align(off + std::max(1, h->n_kv) * (int)it_words * 8);
    lp.off_keymask = off; off = align(off + 3 * (int)d->n_keys * (int)it_words * 8);
    lp.off_allocok = off; off = align(off + (int)it_words * 8);
    lp.off_kvslot = off; off = align(off + (int)req_words * 64 * 2);
    lp.off_tmpl = off; off = align(off + std::max(1u, d->n_templates) * lay.c_hot_words() * 8);
    lp.off_tmplcold = off; off = align(off + std::max(1u, d->n_templates) * lay.cold_words() * 8);
    lp.off_scratch = off; off = align(off + (int)sizeof(ks::Scratch));
    lp.off_dgov = off; off = align(off + std::max(1, P.n_dg) * (int)d->n_res * 8);
    lp.off_dgits = off; off = align(off + std::max(1, P.n_dg) * (int)it_words * 8);
    lp.off_cache = off; off = align(off + 32 * lay.c_hot_words() * 8);
    // The topology groups' descriptors and small mutable state in LDS, when they are few and small enough (round 4 measured 136
    // dependent vector reads per pod on the configs[2] shape, a tenth of them these: 1.08-1.09x, profiles/round4/experiments). Only
    // the one-problem kernels (ksolve_pack / ksolve_pack_big) use the room; a sweep's plan leaves it out (sweep_run).
    lp.off_topo = 0; lp.topo_bytes = 0;
    if (h->has_topology) {
      const ks::TopoView& Tv = P.topo;
      auto al8 = [](size_t b) { return (b + 7) & ~(size_t)7; };
      const size_t G = (size_t)Tv.n_groups, dw = (size_t)Tv.dom_words;
      const size_t tb = al8(G) * 4 + al8(G * 4) * 3 + al8(G * 2) + al8((G + 1) * 4) + al8(G * 8) + al8(G * dw * 8) + al8(G * dw * 64 * 4) + al8(G * 4) + 64;
      if (tb <= 24 * 1024) { lp.off_topo = off; lp.topo_bytes = (int)tb; off = align(off + (int)tb); }
    }
    const int budget = 160 * 1024 - 512 - ks::kSweepLdsExtra;   // (the batched and sweep kernels keep the view and a workspace record behind the plan: ksolve_pack_batch.hip)
    if (off + 13 * 64 > budget) return fail(h, KSOLVE_ERR_UNSUPPORTED, "instance-type tables do not fit the 160 KiB LDS of one CU");
    // the claim order (12 B per claim) and the closed bitmap (1 bit per claim) get what is left
    int cap = (int)(((long long)(budget - off) * 8) / (12 * 8 + 1));
    cap &= ~63;
    if (cap > 8192) cap = 8192;   // two dead-row words per lane in the first-fit scan
    if (h->opts.lds_claim_cap && (int)((h->opts.lds_claim_cap + 63) & ~63u) < cap) cap = (int)((h->opts.lds_claim_cap + 63) & ~63u);
    lp.stage_words = 0; lp.off_stage = off;
    P.big = 0;
    // Problems that turn out to need more in-flight claims than the LDS order holds are re-run with the BIG engine (order
    // in HBM; only the live-set staging words and the closed bitmap, 2 bits per claim, in LDS): plan it now.
    h->lds_big = lp;
    h->big_capable = false;
    if ((int)mc > cap) {
      const long long fit = ((long long)(budget - off - 64 - (int)sizeof(ks::RunTables)) * 8 / 2) & ~63ll;
      if ((long long)mc > fit) { mc = (uint32_t)fit; h->max_claims = mc; h->claim_words = (mc + 63) / 64; W.max_claims = (int)mc; W.claim_words = (int)h->claim_words; }
      ks::LdsPlan& lb = h->lds_big;
      int ob = off;
      lb.order_cap = 0; lb.off_order = ob; ob = align(ob + (int)sizeof(ks::RunTables));   // the claim order's ring tables (run_order.h)
      lb.stage_words = (int)((mc + 63) / 64);
      lb.off_closed = ob; ob = align(ob + lb.stage_words * 8);
      lb.off_stage = ob; ob = align(ob + lb.stage_words * 8);
      lb.total_bytes = ob;
      h->big_capable = (int)mc > cap;
    }
    if (cap > (int)mc) cap = ((int)mc + 63) & ~63;
    lp.off_order = off; lp.order_cap = cap;
    off = align(off + cap * 12);
    lp.off_closed = off; off = align(off + cap / 8 + 8);
    lp.total_bytes = off;
  }
  {
    bool any_minv = false;
    if (d->tmpl_reqs.min_values) for (size_t i = 0; i < (size_t)d->n_templates * d->n_keys; ++i) if (d->tmpl_reqs.min_values[i] >= 0) any_minv = true;
    const bool bounds = any_nonzero(d->pod_reqs.has_gte, d->n_pod_rows) || any_nonzero(d->pod_reqs.has_lte, d->n_pod_rows) ||
                        any_nonzero(d->tmpl_reqs.has_gte, d->n_templates) || any_nonzero(d->tmpl_reqs.has_lte, d->n_templates);
    P.plain = (d->topo.n == 0 && d->n_nodes == 0 && !d->tmpl_daemon_first && !any_minv && !P.reserved_on && !bounds && !d->n_override_groups && !P.hp_on && !P.vol_on) ? 1 : 0;
    P.plain_topo = (d->n_nodes == 0 && !d->tmpl_daemon_first && !any_minv && !P.reserved_on && !bounds && !d->n_override_groups && !P.hp_on && !P.vol_on) ? 1 : 0;
    P.lite = (P.plain && req_words <= 64 && it_words <= 8 && d->n_res <= 4) ? 1 : 0;
#ifdef KSOLVE_NO_LITE
    P.lite = 0;   // A/B builds only
#endif
  }
  {
    // cursor engine (fast_engine.h): candidate when the problem is lite and has no relaxation rows; the kernel itself checks the
    // rest (positive operators only, packed variable keys, 31-bit quantities) and hands the problem back otherwise
    ks::FastWork& fw = h->fw;
    fw.enabled = (P.plain && d->n_res <= 4 && h->opts.engine != 1 && d->n_pod_rows == d->n_pods && d->n_pods > 0) ? 1 : 0;
    if (fw.enabled) {
      h->fast_mc = mc;
      fast_plan_set(h, h->opts.engine == 4 ? 2 : h->opts.engine == 3 ? 1 : 0, 1);
      if ((h->opts.engine == 0 || h->opts.engine == 2) && !h->opts.lds_claim_cap) {
        // Which plan will hold the problem's NodeClaims is predictable from the rows themselves: no packing needs fewer claims than
        // the batch's total requests over the largest allocatable of any type, per resource (round-5 review: the exact configs[3]
        // batch — 27,345 claims — spent a whole attempt on the LDS plan before it moved to plan 2). A problem whose bound already
        // exceeds a plan's capacity starts on the next one; anything the bound misses still moves on by itself (reason 26).
        double need = 0;
        for (uint32_t r = 0; r < d->n_res; ++r) {
          int64_t best = 0;
          for (uint32_t t = 0; t < d->n_its; ++t) best = std::max(best, d->it_allocatable[(size_t)r * d->n_its + t]);
          if (best <= 0) continue;
          long double sum = 0;
          const int64_t* col = d->pod_requests + (size_t)r * d->n_pod_rows;
          for (uint32_t p = 0; p < d->n_pods; ++p) sum += (long double)col[p];
          need = std::max(need, (double)(sum / (long double)best));
        }
        h->fast_need = need;
        int plan = 0;
        while (plan < 2 && need > (double)h->fw.plan.cap) { ++plan; fast_plan_set(h, plan, h->fw.plan.rows); }
      }
      fw.var = dz<ks::FastVar>(h, 1);
      fw.c_hostseq = dz<uint32_t>(h, mc); fw.c_ent = dz<uint16_t>(h, mc);
      fw.c_state = dz<ks::FastClaim>(h, mc); fw.c_npods = dz<uint32_t>(h, mc);
      fw.c_rec = dz<uint64_t>(h, (size_t)mc * (sizeof(ks::FastRec<ks::kFastRows>) / 8));
      fw.max_active = dz<uint32_t>(h, 1);
      fw.ent_its = dz<uint64_t>(h, (size_t)ks::kFastEnt * it_words);
      fw.q_class = dz<uint32_t>(h, d->n_pods); fw.q_claim = dz<uint32_t>(h, d->n_pods); fw.q_cnt = dz<uint32_t>(h, d->n_pods);
      { const size_t oc = std::min<size_t>(65472, ((size_t)mc + 63) & ~(size_t)63) + 64; fw.o_key = dz<uint16_t>(h, oc); fw.o_ord = dz<uint16_t>(h, oc); fw.o_snap = dz<uint16_t>(h, oc); }
      h->d_fast_args = dz<ks::FastArgs>(h, 1);
    }
    // spread engine (topo_engine.h): candidate when the problem is plain but for its topology groups and has no relaxation rows; the
    // kernel checks the rest (which kinds of groups, positive operators, ...) and hands the problem back otherwise. It runs on the
    // cursor engine's tables (requirement-set cache, class records, queue-order arrays) and on the BIG engine's claim order (run_order.h).
    ks::TopoWork& tw = h->tw;
    tw.enabled = (!fw.enabled && P.plain_topo && d->topo.n > 0 && d->topo.n <= (uint32_t)ks::kTopoMaxGroups && d->n_res <= 4 && (h->opts.engine == 0 || h->opts.engine == 6) &&
                  d->n_pod_rows == d->n_pods && d->n_pods > 0 && !d->pod_node) ? 1 : 0;
    if (tw.enabled) {
      fw.var = dz<ks::FastVar>(h, 1);
      fw.c_hostseq = dz<uint32_t>(h, mc); fw.c_ent = dz<uint16_t>(h, mc);
      fw.c_state = dz<ks::FastClaim>(h, mc); fw.c_npods = dz<uint32_t>(h, mc);
      fw.c_rec = nullptr; fw.max_active = dz<uint32_t>(h, 1);
      fw.ent_its = dz<uint64_t>(h, (size_t)ks::kFastEnt * it_words);
      fw.q_class = dz<uint32_t>(h, d->n_pods); fw.q_claim = dz<uint32_t>(h, d->n_pods); fw.q_cnt = dz<uint32_t>(h, d->n_pods);
      fw.o_key = nullptr; fw.o_ord = nullptr; fw.o_snap = nullptr;
      tw.rec = dz<ks::TopoRec>(h, mc);
      auto align = [](int x) { return (x + 15) & ~15; };
      ks::FastPlan& fp = fw.plan;
      int off = 0;
      fp.rows = 1; fp.helper = 0; fp.global_state = 2; fp.cap = 0;
      fp.off_ent = off; off = align(off + ks::kFastEnt * (int)sizeof(ks::FastEnt));
      fp.off_pool = off; off = align(off + ks::kFastPool * 16);
      fp.off_slot = off; off = align(off + 64 * (int)sizeof(ks::FastSlot));
      fp.off_misc = off; off = align(off + (int)sizeof(ks::FastMisc));
      fp.off_hot = off; off = align(off + (int)sizeof(ks::FastHot));
      fp.off_state = fp.off_key = fp.off_ord = fp.off_snap = off;
      tw.plan.off_run = off; off = align(off + (int)sizeof(ks::RunTables));
      tw.plan.off_state = off; off = align(off + (int)sizeof(ks::TopoState));
      tw.plan.total_bytes = off; fp.total_bytes = off;
      h->d_topo_args = dz<ks::TopoArgs>(h, 1);
    }
  }
  be_sync(h);
  be_toc(h, T_UPLOAD);
  if (!be_ok(h)) return fail(h, KSOLVE_ERR_DEVICE, h->error.empty() ? "device allocation/upload failed" : h->error);
  return KSOLVE_OK;
}

static ksolve_status solve_prepare(ksolve_handle* h, bool fresh_context);


// ksolve_probe_create: a probe handle is its descriptor (nodes that are not there, pods to place, NodePool limits); solving it
// — alone or in a batch — goes through sweep_run below, which shares `base`'s device tables.
static ksolve_status probe_create(ksolve_handle* base, const ksolve_probe* pr, ksolve_handle* h) {
  if (!base || base->base || !pr) return fail(h, KSOLVE_ERR_INVALID, "probe of a null handle / of a probe");
  if (base->has_topology && !base->resident) return fail(h, KSOLVE_ERR_UNSUPPORTED, "probes of a problem with topology groups need a resident-cluster base (ksolve_problem_desc.pod_node): its counts include the candidates' pods");
  if (pr->n_pods && !pr->pods) return fail(h, KSOLVE_ERR_INVALID, "probe pods missing");
  if (base->n_nodes && !pr->removed_nodes) return fail(h, KSOLVE_ERR_INVALID, "probe removed_nodes missing");
  h->pr_pods.assign(pr->pods, pr->pods + pr->n_pods);
  for (uint32_t p : h->pr_pods) if (p >= base->n_pods) return fail(h, KSOLVE_ERR_INVALID, "probe pod index out of range");
  {
    std::vector<uint32_t> sorted = h->pr_pods;
    std::sort(sorted.begin(), sorted.end());
    for (size_t i = 1; i < sorted.size(); ++i) if (sorted[i] == sorted[i - 1]) return fail(h, KSOLVE_ERR_INVALID, "probe pod listed twice");
  }
  for (uint32_t e = 0; e < base->n_nodes; ++e) if ((pr->removed_nodes[e >> 6] >> (e & 63)) & 1) h->pr_nodes.push_back(e);
  if (pr->tmpl_limits) h->pr_limits.assign(pr->tmpl_limits, pr->tmpl_limits + (size_t)base->n_templates * (base->n_res + 1));
  h->base = base;
  h->opts = base->opts;
  h->n_keys = base->n_keys; h->req_words = base->req_words; h->n_res = base->n_res; h->n_its = base->n_its; h->it_words = base->it_words;
  h->n_templates = base->n_templates; h->n_pods = base->n_pods; h->n_rows = base->n_rows; h->n_classes = base->n_classes;
  h->n_nodes = base->n_nodes; h->has_topology = base->has_topology;
  h->d_cancel = (int*)be_alloc(h, 4);
  be_sync(h);
  if (!be_ok(h)) return fail(h, KSOLVE_ERR_DEVICE, h->error.empty() ? "device allocation failed" : h->error);
  return KSOLVE_OK;
}

struct ResultsImpl {
  std::vector<int32_t> assign, tmpl;
  std::vector<uint8_t> err, diag, relaxed;
  std::vector<uint32_t> slot, npods, defined, complement, has_gte, has_lte, host_seq, ord;
  std::vector<uint64_t> its, mask, reserved;
  std::vector<int32_t> t_idx; std::vector<uint32_t> t_cnt; std::vector<uint8_t> t_fail;
  std::vector<int64_t> requests, gte, lte;
  std::vector<int32_t> minv;
  std::vector<double> cheapest;
  std::vector<uint32_t> node_npods;
};

// Phases 1-3 (instance-type index, pod classes, queue order) and the resets the pack kernel needs.
// fresh_context: this is the start of a Solve() call (not the re-run of one on the BIG engine): every Solve starts with
// a context that is not cancelled. The flag is cleared synchronously, so a ksolve_cancel that arrives at any later
// moment of the call — during classification, the queue sort or the pack kernel — is kept.
static ksolve_status solve_prepare(ksolve_handle* h, bool fresh_context = true) {
  ks::ProblemView& P = h->pv;
  ks::Workspace& W = h->ws;
  const uint32_t n_pods = h->n_pods, n_rows = h->n_rows, n_res = h->n_res;
  h->engine_used = 1;
  if (fresh_context) {
    be_fill(h, h->d_cancel, 0, 4);
#ifdef KSOLVE_TEST_HOOKS
    if (const char* at = getenv("KSOLVE_TEST_CANCEL_AT")) { const int v = -atoi(at); if (v < 0) be_h2d(h, h->d_cancel, &v, 4); }   // deterministic cancellation for the tests
#endif
    be_sync(h);
  }
  // ---- phase 1: instance-type requirement index ----
  be_tic(h, T_INDEX);
  be_fill(h, (void*)P.kv_has, 0, (size_t)h->req_words * 64 * h->it_words * 8);
  be_fill(h, (void*)P.key_undef, 0, (size_t)h->n_keys * h->it_words * 8);
  be_fill(h, (void*)P.key_compl, 0, (size_t)h->n_keys * h->it_words * 8);
  be_fill(h, (void*)P.key_neg, 0, (size_t)h->n_keys * h->it_words * 8);
  be_fill(h, (void*)P.it_alloc_ok, 0, (size_t)h->it_words * 8);
  be_fill(h, h->it_args.error, 0, 4);
  if (h->n_its) be_launch_it_index(h, (int)h->n_its, h->it_args);
  be_toc(h, T_INDEX);

  // ---- phase 2: pod equivalence classes ----
  be_tic(h, T_CLASSIFY);
  ks::RowArgs& R = h->row_args;
  uint32_t n_classes = 0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    be_fill(h, R.table_hash, 0, (size_t)R.table_size * 8);
    be_fill(h, R.table_rep, 0xFF, (size_t)R.table_size * 4);
    be_fill(h, R.n_classes, 0, 4);
    be_fill(h, R.collision, 0, 4);
    if (n_rows) { be_tic(h, T_ROWHASH); be_launch_row_hash(h, (int)n_rows, R); be_toc(h, T_ROWHASH); }
    uint32_t coll = 0;
    be_d2h(h, &n_classes, R.n_classes, 4);
    be_d2h(h, &coll, R.collision, 4);
    be_sync(h);
    if (!coll) break;
    if (coll == 2) return fail(h, KSOLVE_ERR_DEVICE, "row_diff_far disagrees with rows_equal");   // raised by the test emulation only
    R.seed = R.seed * 6364136223846793005ull + 1442695040888963407ull;  // a 64-bit collision between different rows: re-seed
    if (attempt == 3) return fail(h, KSOLVE_ERR_DEVICE, "row hash collisions persist");
  }
  uint32_t it_err = 0;
  be_d2h(h, &it_err, h->it_args.error, 4);
  be_sync(h);
  if (it_err & 1) return fail(h, KSOLVE_ERR_UNSUPPORTED, "an instance type does not carry `node.kubernetes.io/instance-type In [own name]`");
  if (it_err & 2) return fail(h, KSOLVE_ERR_UNSUPPORTED, "instance types with Gt/Lt requirements are not supported");
  h->n_classes = n_classes;
  if (n_classes > h->class_capacity) {
    // class tables are sized once the class count is known (and kept for later solves on the same handle)
    h->class_capacity = n_classes;
    R.class_rep = dz<uint32_t>(h, n_classes);
    R.cls_requests = dz<int64_t>(h, (size_t)n_classes * n_res);
    h->d_cls_reqs = alloc_reqs(h, n_classes, h->req_words, h->n_keys);
    h->d_cls_strict = alloc_reqs(h, n_classes, h->req_words, h->n_keys);
    R.cls_reqs = h->d_cls_reqs; R.cls_strict = h->d_cls_strict;
    R.cls_tolerates = dz<uint64_t>(h, n_classes);
    R.cls_host_ports = P.hp_on ? dz<uint64_t>(h, (size_t)n_classes * 2) : nullptr;
    R.cls_vol = P.vol_on ? dz<uint64_t>(h, n_classes) : nullptr;
    R.cls_hot = dz<uint64_t>(h, (size_t)n_classes * P.lay.k_hot_words());
    R.cls_cold = dz<uint64_t>(h, (size_t)n_classes * P.lay.cold_words());
    R.cls_topo = h->has_topology ? dz<uint64_t>(h, (size_t)n_classes * 2 * R.topo_words) : nullptr;
    R.lay = P.lay;
    h->ws.dead = dz<uint64_t>(h, (size_t)n_classes * h->claim_words);
    if (h->fw.enabled || h->tw.enabled) { h->fw.cls = dz<ks::FastSlot>(h, n_classes); h->fw.cls_first = dz<uint32_t>(h, n_classes); h->fw.cls_last = dz<uint32_t>(h, n_classes); }
    if (h->tw.enabled) h->tw.cls = dz<ks::TopoClass>(h, n_classes);
    if (h->n_nodes) h->ws.n_dead = dz<uint64_t>(h, (size_t)n_classes * P.node_words);
  }
  if (n_classes > 0) {
    be_fill(h, R.min_request, 0x7F, (size_t)n_res * 8);
    be_launch_row_class(h, (int)n_rows, R);
    be_launch_class_gather(h, (int)n_classes, R);
  }
  P.n_classes = (int)n_classes;
  P.cls_requests = R.cls_requests; P.cls_reqs = as_const(h->d_cls_reqs); P.cls_strict = as_const(h->d_cls_strict);
  P.cls_tolerates = R.cls_tolerates;
  P.cls_hp = R.cls_host_ports;
  P.cls_vol = R.cls_vol;
  P.cls_hot = R.cls_hot; P.cls_cold = R.cls_cold;
  P.topo.cls_topo = R.cls_topo;
  if (n_classes) be_fill(h, W.dead, 0, (size_t)n_classes * h->claim_words * 8);
  if (n_classes && h->n_nodes) be_fill(h, W.n_dead, 0, (size_t)n_classes * P.node_words * 8);
  be_toc(h, T_CLASSIFY);

  // ---- phase 3: queue order ----
  be_tic(h, T_SORT);
  be_sort_pods(h);
  if (h->fw.enabled && !P.big && n_pods && n_classes) {
    // the cursor engine reads the queue's classes in queue order; how many classes are live at once decides its rows of class slots
    be_fill(h, h->fw.cls_first, 0xFF, (size_t)n_classes * 4); be_fill(h, h->fw.cls_last, 0, (size_t)n_classes * 4); be_fill(h, h->fw.max_active, 0, 4);
    // (the count of classes live at once — quadratic in the classes, and a blocking 4-byte download — belongs to the problem, not to
    // the solve: a handle's rows do not change, so later solves of the handle reuse the first one's. ADVICE r5.)
    const bool count_live = !h->fast_live_known || h->fast_live_classes != n_classes;
    be_launch_fast_queue(h, count_live);
    uint32_t live = h->fast_live;
    if (count_live) {
      live = 0xFFFFFFFFu;
      if (n_classes <= 64) live = n_classes;
      else if (n_classes <= 32768) { be_d2h(h, &live, h->fw.max_active, 4); be_sync(h); }   // (beyond: four rows)
      h->fast_live = live; h->fast_live_known = true; h->fast_live_classes = n_classes;
    }
    const int rows = live <= 64 ? 1 : ks::kFastRows;
    if (rows != h->fw.plan.rows) {
      fast_plan_set(h, h->fw.plan.global_state, rows);
      if ((h->opts.engine == 0 || h->opts.engine == 2) && !h->opts.lds_claim_cap)   // (a plan's capacity depends on the rows of class slots: the bound of create() once more)
        while (h->fw.plan.global_state < 2 && h->fast_need > (double)h->fw.plan.cap) fast_plan_set(h, h->fw.plan.global_state + 1, rows);
    }
  }
  if (h->tw.enabled && n_pods && n_classes) {
    // the spread engine reads the queue's classes in queue order too (no class slots: the overlap count is not needed)
    be_fill(h, h->fw.cls_first, 0xFF, (size_t)n_classes * 4); be_fill(h, h->fw.cls_last, 0, (size_t)n_classes * 4); be_fill(h, h->fw.max_active, 0, 4);
    be_launch_fast_queue(h, false);
  }
  be_toc(h, T_SORT);

  // ---- phase 4: pack ----
  be_fill(h, W.last_len, 0, (size_t)n_pods * 4);
  be_fill(h, W.assign, 0xFF, (size_t)n_pods * 4);
  be_fill(h, W.err, 0, n_pods); be_fill(h, W.diag, 0, n_pods);
  be_fill(h, W.n_claims_out, 0, 4); be_fill(h, W.status_out, 0, 4);
  if (P.topo.n_host_groups) be_fill(h, W.tg_claim_counts, 0, (size_t)P.topo.n_host_groups * h->max_claims * 4);
  if (P.topo.n_host_groups) be_fill(h, W.host_le, 0xFF, (size_t)P.topo.n_host_groups * 2 * h->claim_words * 8);
  if (P.topo.n_key_slots) be_fill(h, W.kv_claims, 0, (size_t)P.topo.n_key_slots * 64 * h->claim_words * 8);
  return KSOLVE_OK;
}

struct ClaimCols {   // the columns of ksolve_claims while they are being decoded from claim records
  std::vector<int32_t> tmpl, minv;
  std::vector<uint32_t> npods, defined, complement, has_gte, has_lte, host_seq;
  std::vector<uint8_t> relaxed;
  std::vector<uint64_t> its, mask;
  std::vector<int64_t> requests, gte, lte;
  void resize(size_t C, const ksolve_handle* h) {
    tmpl.resize(C); npods.resize(C); defined.resize(C); complement.resize(C); has_gte.resize(C); has_lte.resize(C); host_seq.resize(C); relaxed.resize(C);
    its.resize(C * h->it_words); mask.resize(C * h->req_words); requests.resize(C * h->n_res); gte.resize(C * h->n_keys); lte.resize(C * h->n_keys); minv.resize(C * h->n_keys);
  }
};
// one hot / cold claim record (RecLayout) -> row c of the columns; `reserved` = reservation ids the claim holds, `dreq` = addDaemonRequests
static void decode_claim(const ksolve_handle* h, const ks::ProblemView& P, uint32_t c, const uint64_t* hr, const uint64_t* cr, uint64_t reserved, const int64_t* dreq, ClaimCols& o) {
  const ks::RecLayout ly = P.lay;
  const uint32_t n_res = h->n_res;
  auto &tmpl = o.tmpl; auto &npods = o.npods, &defined = o.defined, &complement = o.complement, &has_gte = o.has_gte, &has_lte = o.has_lte, &host_seq = o.host_seq;
  auto &relaxed = o.relaxed; auto &its = o.its, &mask = o.mask; auto &requests = o.requests, &gte = o.gte, &lte = o.lte; auto &minv = o.minv;
    for (uint32_t x = 0; x < h->req_words; ++x) mask[(size_t)c * h->req_words + x] = hr[ly.c_mask() + x];
    for (uint32_t x = 0; x < h->it_words; ++x) its[(size_t)c * h->it_words + x] = hr[ly.c_its() + x];
    for (uint32_t r = 0; r < n_res; ++r) requests[(size_t)c * n_res + r] = (int64_t)hr[ly.c_total() + r] + dreq[r];   // FinalizeScheduling, nodeclaim.go:405-408
    defined[c] = (uint32_t)hr[ly.c_f0()]; complement[c] = (uint32_t)(hr[ly.c_f0()] >> 32);
    has_gte[c] = (uint32_t)hr[ly.c_f1()]; has_lte[c] = (uint32_t)(hr[ly.c_f1()] >> 32);
    if (P.reserved_on && reserved) {
      // FinalizeScheduling pins the claim to its reservations (nodeclaim.go:391-403): capacity-type = reserved,
      // reservation-id In [held ids]
      const uint32_t kc = (uint32_t)P.dict.key_ct, kr = (uint32_t)P.key_rid;
      uint64_t* mk = mask.data() + (size_t)c * h->req_words;
      for (uint32_t x = P.dict.key_word_off[kc]; x < P.dict.key_word_off[kc + 1]; ++x) mk[x] = 0;
      mk[P.dict.key_word_off[kc] + (uint32_t)P.ct_reserved / 64] = 1ull << (P.ct_reserved % 64);
      defined[c] |= 1u << kc; complement[c] &= ~(1u << kc); has_gte[c] &= ~(1u << kc); has_lte[c] &= ~(1u << kc);
      const uint32_t rx = P.dict.key_word_off[kr];
      if ((defined[c] >> kr) & 1) mk[rx] = ((complement[c] >> kr) & 1) ? (reserved & ~mk[rx]) : (reserved & mk[rx]);
      else mk[rx] = reserved;
      defined[c] |= 1u << kr; complement[c] &= ~(1u << kr);
    }
    tmpl[c] = (int32_t)(uint32_t)hr[ly.c_meta()]; npods[c] = (uint32_t)(hr[ly.c_meta()] >> 32);
    host_seq[c] = (uint32_t)hr[ly.c_meta2()];
    const uint32_t fl = (uint32_t)(hr[ly.c_meta2()] >> 32);
    relaxed[c] = fl & 1u;
    const bool cold_valid = hr[ly.c_f1()] != 0 || (fl & 2u);
    for (uint32_t k = 0; k < h->n_keys; ++k) {
      gte[(size_t)c * h->n_keys + k] = cold_valid ? ((const int64_t*)cr)[k] : 0;
      lte[(size_t)c * h->n_keys + k] = cold_valid ? ((const int64_t*)cr)[h->n_keys + k] : 0;
      minv[(size_t)c * h->n_keys + k] = (fl & 2u) ? ((const int32_t*)(cr + 2 * h->n_keys))[k] : -1;
    }
}

// Phases 5-6 (finalize, download) after the pack kernel has run.
static ksolve_status solve_finish(ksolve_handle* h, ksolve_results* out) {
  memset(out, 0, sizeof(*out));
  ks::ProblemView& P = h->pv;
  ks::Workspace& W = h->ws;
  const uint32_t n_pods = h->n_pods, n_res = h->n_res;
  int n_claims = 0, status = 0;
  be_d2h(h, &n_claims, W.n_claims_out, 4);
  be_d2h(h, &status, W.status_out, 4);
  be_sync(h);
  if (!be_ok(h)) return fail(h, KSOLVE_ERR_DEVICE, h->error.empty() ? "pack kernel failed" : h->error);
  if (status == 1) return fail(h, KSOLVE_ERR_CAPACITY, "more in-flight NodeClaims than this build keeps resident (ksolve_options.max_claims / up to 8192 per problem, bounded by the 160 KiB LDS)");

  // ---- phase 5: finalize ----
  be_tic(h, T_FINALIZE);
  ks::FinalizeArgs F{P.dict, (int)h->n_its, (int)h->it_words, P.n_zones, P.n_cts, P.it_off_avail, P.it_off_price, W.c_hot, W.c_cold, P.lay, h->d_cheapest,
                     P.dg_first, P.dg_ov, P.dg_its, P.dg_nonempty, W.t_its, h->d_daemon_requests,
                     P.reserved_on ? W.c_reserved : nullptr, P.it_resv_first, P.resv_zone, P.resv_id, P.resv_price,
                     0, 0, P.it_reqs, nullptr, nullptr, nullptr, nullptr};
  if (h->opts.truncate_instance_types && n_claims) {
    if ((size_t)n_claims > h->trunc_capacity) {
      h->trunc_capacity = (size_t)n_claims;
      h->d_sort_idx = dz<int32_t>(h, (size_t)n_claims * h->n_its); h->d_sort_price = dz<double>(h, (size_t)n_claims * h->n_its);
      h->d_ordered_count = dz<uint32_t>(h, n_claims); h->d_trunc_failed = dz<uint8_t>(h, n_claims);
    }
    F.truncate_n = (int)h->opts.truncate_instance_types; F.best_effort = h->opts.min_values_best_effort ? 1 : 0;
    F.sort_idx = h->d_sort_idx; F.sort_price = h->d_sort_price; F.ordered_count = h->d_ordered_count; F.trunc_failed = h->d_trunc_failed;
  }
  if (n_claims) be_launch_finalize(h, n_claims, F);
  be_toc(h, T_FINALIZE);

  // ---- phase 6: download ----
  be_tic(h, T_DOWNLOAD);
  ResultsImpl* im = new ResultsImpl();
  im->assign.resize(n_pods); im->err.resize(n_pods); im->diag.resize(n_pods); im->slot.resize(n_pods);
  if (n_pods) {
    be_d2h(h, im->assign.data(), W.assign, (size_t)n_pods * 4);
    be_d2h(h, im->err.data(), W.err, n_pods); be_d2h(h, im->diag.data(), W.diag, n_pods);
    be_d2h(h, im->slot.data(), W.slot, (size_t)n_pods * 4);
  }
  const uint32_t C = (uint32_t)n_claims;
  const ks::RecLayout ly = P.lay;
  ClaimCols cc; cc.resize(C, h);
  auto &tmpl = cc.tmpl; auto &npods = cc.npods, &defined = cc.defined, &complement = cc.complement, &has_gte = cc.has_gte, &has_lte = cc.has_lte, &host_seq = cc.host_seq;
  auto &relaxed = cc.relaxed; auto &its = cc.its, &mask = cc.mask; auto &requests = cc.requests, &gte = cc.gte, &lte = cc.lte; auto &minv = cc.minv;
  std::vector<uint32_t> ord(C);
  std::vector<double> cheapest(C);
  std::vector<int64_t> daemon_req;
  std::vector<uint64_t> reserved;
  std::vector<int32_t> t_idx; std::vector<uint32_t> t_cnt; std::vector<uint8_t> t_fail;
  std::vector<uint64_t> hot((size_t)C * ly.c_hot_words()), cold((size_t)C * ly.cold_words());
  if (C) {
    be_d2h(h, hot.data(), W.c_hot, hot.size() * 8);
    be_d2h(h, cold.data(), W.c_cold, cold.size() * 8);
    be_d2h(h, cheapest.data(), h->d_cheapest, (size_t)C * 8);
    reserved.resize(C);
    be_d2h(h, reserved.data(), W.c_reserved, (size_t)C * 8);
    if (h->opts.truncate_instance_types) {
      t_idx.resize((size_t)C * h->n_its); t_cnt.resize(C); t_fail.resize(C);
      be_d2h(h, t_idx.data(), h->d_sort_idx, t_idx.size() * 4);
      be_d2h(h, t_cnt.data(), h->d_ordered_count, (size_t)C * 4);
      be_d2h(h, t_fail.data(), h->d_trunc_failed, C);
    }
    daemon_req.resize((size_t)C * n_res);
    be_d2h(h, daemon_req.data(), h->d_daemon_requests, (size_t)C * n_res * 8);
    be_d2h(h, ord.data(), W.o_ord, (size_t)C * 4);
  }
  im->node_npods.resize(h->n_nodes);
  if (h->n_nodes) be_d2h(h, im->node_npods.data(), W.n_npods, (size_t)h->n_nodes * 4);
  ks::Counters ctr{};
  be_d2h(h, &ctr, W.counters, sizeof(ctr));
  be_sync(h);
  be_toc(h, T_DOWNLOAD);
  if (!be_ok(h)) { delete im; return fail(h, KSOLVE_ERR_DEVICE, h->error.empty() ? "download failed" : h->error); }
  for (uint32_t c = 0; c < C; ++c)
    decode_claim(h, P, c, hot.data() + (size_t)c * ly.c_hot_words(), cold.data() + (size_t)c * ly.cold_words(), reserved[c], daemon_req.data() + (size_t)c * n_res, cc);

  // Report claims in the order the reference's s.newNodeClaims slice ends in (position order), so claim index i in the
  // results is position i; pod assignments are remapped accordingly.
  std::vector<uint32_t> newidx(C);
  for (uint32_t i = 0; i < C; ++i) newidx[ord[i]] = i;
  auto permute = [&](auto& dst, const auto& src, size_t width) {
    dst.resize((size_t)C * width);
    for (uint32_t i = 0; i < C; ++i) for (size_t w = 0; w < width; ++w) dst[(size_t)i * width + w] = src[(size_t)ord[i] * width + w];
  };
  permute(im->tmpl, tmpl, 1); permute(im->npods, npods, 1); permute(im->its, its, h->it_words); permute(im->mask, mask, h->req_words);
  permute(im->defined, defined, 1); permute(im->complement, complement, 1); permute(im->has_gte, has_gte, 1); permute(im->has_lte, has_lte, 1);
  permute(im->gte, gte, h->n_keys); permute(im->lte, lte, h->n_keys); permute(im->minv, minv, h->n_keys);
  if (reserved.empty()) reserved.assign(C, 0);
  permute(im->reserved, reserved, 1);
  if (!t_cnt.empty()) { permute(im->t_idx, t_idx, h->n_its); permute(im->t_cnt, t_cnt, 1); permute(im->t_fail, t_fail, 1); }
  permute(im->requests, requests, n_res); permute(im->host_seq, host_seq, 1); permute(im->relaxed, relaxed, 1); permute(im->cheapest, cheapest, 1);
  for (uint32_t p = 0; p < n_pods; ++p) if (im->assign[p] >= 0) im->assign[p] = (int32_t)newidx[im->assign[p]];
  double cost = 0;
  for (uint32_t i = 0; i < C; ++i) if (im->cheapest[i] < 1e300) cost += im->cheapest[i];

  out->status = status == 2 ? KSOLVE_ERR_CANCELLED : KSOLVE_OK;
  out->n_pods = n_pods;
  out->pod_assignment = im->assign.data(); out->pod_error = im->err.data(); out->pod_error_diag = im->diag.data(); out->pod_slot = im->slot.data();
  ksolve_claims& cl = out->claims;
  cl.n_claims = C; cl.it_words = h->it_words; cl.req_words = h->req_words; cl.n_keys = h->n_keys; cl.n_res = n_res;
  cl.template_idx = im->tmpl.data(); cl.pod_count = im->npods.data(); cl.it_mask = im->its.data(); cl.requests = im->requests.data();
  cl.req_mask = im->mask.data(); cl.req_defined = im->defined.data(); cl.req_complement = im->complement.data();
  cl.req_has_gte = im->has_gte.data(); cl.req_has_lte = im->has_lte.data(); cl.req_gte = im->gte.data(); cl.req_lte = im->lte.data();
  cl.req_min_values = im->minv.data(); cl.min_values_relaxed = im->relaxed.data(); cl.cheapest_price = im->cheapest.data();
  cl.hostname_seq = im->host_seq.data(); cl.reserved_mask = im->reserved.data();
  cl.n_instance_types = h->n_its;
  if (!im->t_cnt.empty()) { cl.ordered_instance_types = im->t_idx.data(); cl.ordered_count = im->t_cnt.data(); cl.truncation_failed = im->t_fail.data(); }
  out->bin_evaluations = ctr.bin_evaluations; out->it_evaluations = ctr.it_evaluations; out->queue_pops = ctr.queue_pops;
  out->sorts = ctr.sorts; out->slow_sorts = ctr.slow_sorts; out->relaxations = ctr.relaxations;
  out->ref_bin_evaluations = ctr.ref_bin_evaluations;
  for (int i = 0; i < 24; ++i) out->phase_cycles[i] = ctr.cycles[i];
  out->phase_cycles[23] = ctr.full_filters; out->phase_cycles[22] = ctr.column_resets; out->phase_cycles[21] = ctr.full_evaluations;
  out->us_upload = h->timers.ms[T_UPLOAD] * 1e3;
  out->us_prepass = (h->timers.ms[T_INDEX] + h->timers.ms[T_CLASSIFY] + h->timers.ms[T_SORT]) * 1e3;
  out->us_pack = h->timers.ms[T_PACK] * 1e3; out->us_finalize = h->timers.ms[T_FINALIZE] * 1e3; out->us_download = h->timers.ms[T_DOWNLOAD] * 1e3;
  out->packing_cost = cost;
  out->engine_used = h->engine_used; out->engine_fallback_reason = h->fast_reason; out->cursor_wide = h->engine_used == 2 ? (uint32_t)h->fw.plan.global_state : 0u; out->cursor_attempts = h->fast_attempts;
  out->impl = im;
  return out->status;
}

// BIG engine: one ring per pod count for the claim order (run_order.h). A run of claims with k pods each holds at most
// n_pods / k claims (and never more than max_claims): ring k gets the next power of two.
static void alloc_run_order(ksolve_handle* h) {
  if (h->ws.o_ring) return;
  const size_t kmax = (size_t)h->n_pods + 2;
  std::vector<uint32_t> off(kmax, 0);
  std::vector<uint8_t> lg(kmax, 0);
  uint64_t total = 0;
  for (size_t k = 1; k < kmax; ++k) {
    uint64_t need = std::min<uint64_t>(h->max_claims, (uint64_t)h->n_pods / (uint64_t)k + 2) + 2;
    int l2 = 1;
    while ((1ull << l2) < need) ++l2;
    off[k] = (uint32_t)total; lg[k] = (uint8_t)l2;
    total += 1ull << l2;
  }
  h->ws.o_ring = dz<uint32_t>(h, (size_t)total);
  h->ws.o_cnt = dz<uint32_t>(h, h->max_claims);
  h->ws.run_tabs = dz<uint32_t>(h, 3 * kmax);
  h->ws.run_off = up(h, off.data(), off.size());
  h->ws.run_log = up(h, lg.data(), lg.size());
  h->ws.run_kmax = (int)kmax;
}
// ---------------------------------------------------------------------------------------------------------------------------
// Sweeps over a resident cluster (ksolve_sweep; probe handles go through the same code). SimulateScheduling
// (disruption/helpers.go:53-155) runs Solve() once per candidate set against the same cluster: the base handle holds the
// cluster (every node an existing node, every displaceable pod a pod row, classes and queue order computed once), a probe is
// (nodes that are not there, pods to place, NodePool limits) and owns only what it creates: its NodeClaims and an overlay of
// the nodes it commits pods to.
struct SweepImpl {   // host side of ksolve_sweep_results
  std::vector<int32_t> status, assign;
  std::vector<uint8_t> err, diag;
  std::vector<uint32_t> slot, claim_off;
  std::vector<uint64_t> ref;
  ResultsImpl claims;
  std::vector<ks::Counters> counters;
};

// base-level tables every probe shares; computed once, after the classes and the queue order exist
static ksolve_status sweep_prepare_base(ksolve_handle* base) {
  if (base->sweep_ready) return KSOLVE_OK;
  if (!base->prepared) {
    ksolve_status st = solve_prepare(base, true);
    if (st != KSOLVE_OK) return st;
    std::vector<uint32_t> order(base->n_pods);
    if (base->n_pods) be_d2h(base, order.data(), base->pv.sorted_pods, (size_t)base->n_pods * 4);
    be_sync(base);
    if (!be_ok(base)) return fail(base, KSOLVE_ERR_DEVICE, base->error);
    base->h_rank.assign(base->n_pods, 0);
    for (uint32_t i = 0; i < base->n_pods; ++i) base->h_rank[order[i]] = i;
    base->prepared = true;
  }
  ks::ProblemView& P = base->pv;
  const uint32_t ne = base->n_nodes, nw = (uint32_t)P.node_words, nc = base->n_classes;
  if (ne) {
    // consolidateAfter bitmap + prefix counts (scheduler.go:628)
    std::vector<uint8_t> fl(ne);
    be_d2h(base, fl.data(), P.node_flags, ne);
    be_sync(base);
    std::vector<uint64_t> skip(nw, 0);
    std::vector<uint32_t> prefix(nw + 1, 0);
    bool any = false;
    for (uint32_t e = 0; e < ne; ++e) if (fl[e] & 2) { skip[e >> 6] |= 1ull << (e & 63); any = true; }
    for (uint32_t w = 0; w < nw; ++w) prefix[w + 1] = prefix[w] + (uint32_t)__builtin_popcountll(skip[w]);
    if (any) { P.node_skip = up(base, skip.data(), skip.size()); P.node_skip_prefix = up(base, prefix.data(), prefix.size()); }
    else { P.node_skip = nullptr; P.node_skip_prefix = nullptr; }
    // every class against every pristine node
    uint64_t* dead0 = dz<uint64_t>(base, (size_t)std::max(1u, nc) * nw);
    ks::NodeDeadArgs a{};
    a.dict = P.dict; a.lay = P.lay; a.n_nodes = (int)ne; a.node_words = (int)nw; a.n_classes = (int)nc; a.hp_on = P.hp_on;
    a.cls_hot = P.cls_hot; a.cls_cold = P.cls_cold; a.cls_hp = P.cls_hp; a.node_taints = P.node_taints;
    a.pristine.mask = base->ws.n_mask0; a.pristine.defined = base->ws.n_defined0; a.pristine.complement = base->ws.n_complement0;
    a.pristine.hg = nullptr; a.pristine.hl = nullptr; a.pristine.gte = nullptr; a.pristine.lte = nullptr;
    a.pristine.remaining = base->ws.n_remaining0; a.pristine.hp = P.node_hp0; a.pristine.stride = ne;
    a.dead0 = dead0;
    if (nc) { be_tic(base, T_INDEX); be_launch_node_dead0(base, (int)nw, a); be_toc(base, T_INDEX); base->dead0_us = base->timers.ms[T_INDEX] * 1e3; }
    P.n_dead0 = dead0;
  }
  base->d_pv = dz<ks::ProblemView>(base, 1);
  be_sync(base);
  if (!be_ok(base)) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "sweep tables failed" : base->error);
  base->sweep_ready = true;
  return KSOLVE_OK;
}

static char* sweep_buffer(ksolve_handle* base, char*& buf, size_t& have, size_t need, bool may_refuse = false) {
  if (need > have) {
    if (buf) be_free(base, buf);
    const size_t want = need + need / 4 + 4096;
    // the arena of a launch may be more than the device has left: that is not an error of the handle (hip_check made the
    // refusal sticky and the retry with smaller launches could never run) — null, and sweep() goes on with half the probes per launch
    buf = (char*)(may_refuse ? be_try_alloc(base, want) : be_alloc(base, want));
    if (!buf && may_refuse) { buf = (char*)be_try_alloc(base, need + 4096); have = buf ? need + 4096 : 0; base->sweep_arena_refused = buf == nullptr; return buf; }   // (once more without the growth margin)
    have = buf ? want : 0;
  }
  return buf;
}

// Runs n probes of `base` in one launch. node_off/nodes, pod_off/pods: CSR descriptors (pods in the caller's order);
// limits: null or n * T*(nr+1); cancel[i]: the flag probe i polls (null entries: the base's own).
static ksolve_status sweep_run(ksolve_handle* base, uint32_t n, const uint32_t* node_off, const uint32_t* nodes, const uint32_t* pod_off, const uint32_t* pods,
                               const int64_t* const* limits, int* const* cancel, SweepImpl* im, double* us) {
  ksolve_status st = sweep_prepare_base(base);
  if (st != KSOLVE_OK) return st;
  ks::ProblemView& P = base->pv;
  const ks::RecLayout lay = P.lay;
  const uint32_t ne = base->n_nodes, nr = base->n_res, T = base->n_templates, nc = std::max(1u, base->n_classes), nk = base->n_keys;
  const uint32_t total_pods = pod_off[n], total_nodes = node_off[n];
  const bool bounds = base->ws.n_hg != nullptr;
  be_tic(base, T_UPLOAD);
  PhaseRange upload_range{base, true};
#ifdef KSOLVE_TEST_HOOKS
  const bool trace = getenv("KSOLVE_TEST_SWEEP_TRACE") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
#endif
  // ---- validate + queue order per probe (the base queue order, restricted) ----
  // what the launch reads — every probe's workspace record, its pods in queue order, its removed nodes — is written straight into
  // the handle's page-locked staging memory: one DMA each instead of a staged copy out of pageable vectors
  // (round 6: the workspace records are laid out by the device from one 48-byte descriptor per probe — kernels.h sweep_item_fill)
  bool any_cancel = false;
  for (uint32_t p = 0; p < n; ++p) any_cancel = any_cancel || (cancel && cancel[p]);
  const size_t st_items = ((size_t)n * sizeof(ks::SweepItemDesc) + 63) & ~(size_t)63, st_sorted = ((size_t)total_pods * 4 + 67) & ~(size_t)63, st_removed = ((size_t)total_nodes * 4 + 67) & ~(size_t)63;
  const size_t st_order = ((size_t)n * 4 + 67) & ~(size_t)63;   // the order the compact launch hands the probes out in
  const size_t st_cancel = any_cancel ? ((size_t)n * 8 + 63) & ~(size_t)63 : 0;
  char* stage = (char*)be_stage(base, st_items + st_sorted + st_removed + st_order + st_cancel);
  if (!stage) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "host staging allocation failed" : base->error);
  ks::SweepItemDesc* const descs = (ks::SweepItemDesc*)stage;
  uint32_t* const sorted = (uint32_t*)(stage + st_items);
  uint32_t* const removed = (uint32_t*)(stage + st_items + st_sorted);
  uint32_t* const ord = (uint32_t*)(stage + st_items + st_sorted + st_removed);
  volatile int** const cancel_h = any_cancel ? (volatile int**)(stage + st_items + st_sorted + st_removed + st_order) : nullptr;
  std::vector<uint32_t> perm(total_pods);   // perm: position in the probe's sorted list -> position in the caller's list
  if (total_nodes) memcpy(removed, nodes, (size_t)total_nodes * 4);
  {
    // every probe on its own (a few host threads): its pods in queue order — (rank << 32 | position) keys, one plain sort — and
    // its removed nodes ascending
    const uint32_t nt = n < 1024 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<const char*> bad(nt, nullptr);
    auto work = [&](uint32_t t) {
      std::vector<uint64_t> key;
      for (uint32_t p = (uint32_t)((uint64_t)n * t / nt), pe = (uint32_t)((uint64_t)n * (t + 1) / nt); p < pe && !bad[t]; ++p) {
        const uint32_t b = pod_off[p], m = pod_off[p + 1] - b;
        key.resize(m);
        for (uint32_t i = 0; i < m; ++i) {
          if (pods[b + i] >= base->n_pods) { bad[t] = "probe pod index out of range"; break; }
          key[i] = ((uint64_t)base->h_rank[pods[b + i]] << 32) | i;
        }
        if (bad[t]) break;
        std::sort(key.begin(), key.end());
        for (uint32_t i = 0; i < m; ++i) { perm[b + i] = (uint32_t)key[i]; sorted[b + i] = pods[b + (uint32_t)key[i]]; }
        for (uint32_t i = 1; i < m; ++i) if (sorted[b + i] == sorted[b + i - 1]) { bad[t] = "probe pod listed twice"; break; }
        std::sort(removed + node_off[p], removed + node_off[p + 1]);
        for (uint32_t i = node_off[p]; i < node_off[p + 1]; ++i) {
          if (removed[i] >= ne) { bad[t] = "probe node index out of range"; break; }
          // a node listed twice would be taken out of the evaluation counts and the topology registrations twice
          if (i > node_off[p] && removed[i] == removed[i - 1]) { bad[t] = "probe node listed twice"; break; }
        }
      }
    };
    if (nt == 1) work(0);
    else { std::vector<std::thread> pool; for (uint32_t t = 0; t < nt; ++t) pool.emplace_back(work, t); for (auto& th : pool) th.join(); }
    for (uint32_t t = 0; t < nt; ++t) if (bad[t]) return fail(base, KSOLVE_ERR_INVALID, bad[t]);
  }
  // ---- LDS plan of the launch: the base plan with the claim order cut down to the largest probe ----
  uint32_t max_m = 1;
  for (uint32_t p = 0; p < n; ++p) max_m = std::max(max_m, pod_off[p + 1] - pod_off[p]);
  ks::LdsPlan lp = P.lds;
  {
    auto align = [](int x) { return (x + 15) & ~15; };
    uint32_t want = max_m;
    if (base->opts.max_claims && base->opts.max_claims < want) want = base->opts.max_claims;
    int cap = lp.order_cap;
    if (cap > (int)((want + 63) & ~63u)) cap = (int)((want + 63) & ~63u);
    int o = lp.topo_bytes ? lp.off_topo : lp.off_order;   // (probes read the topology tables from HBM: no room for them in every workgroup's LDS)
    lp.off_order = o; lp.off_stage = o; lp.off_topo = 0; lp.topo_bytes = 0;
    lp.order_cap = cap;
    o = align(o + cap * 12);
    lp.off_closed = o; o = align(o + cap / 8 + 8);
    lp.total_bytes = o;
    lp.waves = 0; lp.wave_stride = 0; lp.off_shared_misc = 0;
    // The compact form of the launch (ksolve_pack_sweep4, engine.h ScratchSmall): four wavefronts per workgroup share the read-only
    // tables and the template records; each keeps its own working set, a third of the general Scratch. With 256 VGPRs per wavefront
    // (two per SIMD) a CU then runs eight probes at a time instead of four — the launch is bound by the latency of each probe's
    // dependent steps, so the probes in flight are its throughput. Needs dictionaries of <= 32 mask words and <= 512 instance types
    // (ScratchSmall) and a claim order short enough for four working sets beside the shared tables.
    const ks::RecLayout& ly = P.lay;
    bool compact = ly.rw <= ks::ScratchSmall::kReqWords && ly.iw <= ks::ScratchSmall::kItWords;
#ifdef KSOLVE_TEST_HOOKS
    if (getenv("KSOLVE_TEST_SWEEP_GENERAL")) compact = false;   // tests / A-B runs: the one-wavefront kernel with the general Scratch
#endif
    if (compact) {
      ks::LdsPlan c = lp;
      const int np = ly.iw * 64, T = std::max(1, P.n_templates);
      int so = 0;
      c.off_alloc = so; so = align(so + ly.nr * np * 8);
      c.off_avail = so; so = align(so + np * 8);
      c.off_kv = so; so = align(so + std::max(1, lp.n_kv) * ly.iw * 8);
      c.off_keymask = so; so = align(so + 3 * ly.nk * ly.iw * 8);
      c.off_allocok = so; so = align(so + ly.iw * 8);
      c.off_kvslot = so; so = align(so + ly.rw * 64 * 2);
      c.off_tmpl = so; so = align(so + T * ly.c_hot_words() * 8);
      c.off_tmplcold = so; so = align(so + T * ly.cold_words() * 8);
      c.off_dgov = so; so = align(so + std::max(1, P.n_dg) * ly.nr * 8);
      c.off_dgits = so; so = align(so + std::max(1, P.n_dg) * ly.iw * 8);
      c.off_shared_misc = so; so = align(so + 16);
      const int w0 = so;   // wave 0's working set starts here
      c.off_scratch = so; so = align(so + (int)sizeof(ks::ScratchSmall));
      c.off_cache = so; so = align(so + ks::ScratchSmall::kCacheLines * ly.c_hot_words() * 8);
      c.off_order = so; so = align(so + cap * 12);
      c.off_closed = so; so = align(so + cap / 8 + 8);
      c.off_stage = so; c.stage_words = 0;
      c.waves = 4; c.wave_stride = so - w0;
      c.total_bytes = w0 + c.waves * c.wave_stride;
      if (c.total_bytes + ks::kSweepLdsExtra + 16 <= 160 * 1024 - 512) lp = c;
    }
  }
  // ---- arena: every probe's workspace, carved in two passes (measure, then assign) ----
  std::vector<uint32_t> claim_base(n + 1, 0);
  for (uint32_t p = 0; p < n; ++p) {
    uint32_t mc = std::max(1u, pod_off[p + 1] - pod_off[p]);
    if (base->opts.max_claims && base->opts.max_claims < mc) mc = base->opts.max_claims;
    claim_base[p + 1] = claim_base[p] + mc;
  }
  const uint32_t total_mc = claim_base[n];
  char* arena = nullptr;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* q = arena ? (void*)(arena + off) : nullptr; off += (bytes + 63) & ~(size_t)63; return q; };
  // regions shared by all probes (indexed by pod offset / claim slot / probe)
  uint32_t* d_sorted = nullptr; uint32_t* d_removed = nullptr; int64_t* d_limits = nullptr;
  int32_t* d_assign = nullptr; uint32_t* d_slot = nullptr; uint8_t* d_err = nullptr; uint8_t* d_diag = nullptr; uint32_t* d_last = nullptr; uint32_t* d_queue = nullptr;
  uint64_t* d_hot = nullptr; uint64_t* d_cold = nullptr; uint64_t* d_resv = nullptr; uint64_t* d_chp = nullptr; uint32_t* d_okey = nullptr; uint32_t* d_oord = nullptr; uint32_t* d_opos = nullptr;
  int* d_nclaims = nullptr; int* d_status = nullptr; ks::Counters* d_ctr = nullptr; ks::Workspace* d_items = nullptr;
  ks::SweepItemDesc* d_descs = nullptr; volatile int** d_cancel_each = nullptr;
  uint32_t* d_order = nullptr; uint32_t* d_next = nullptr;   // the compact launch hands out probes largest first through a counter (below)
  size_t zero_from = 0, zero_to = 0, ones_to = 0;
  bool any_limits = false;
  for (uint32_t p = 0; p < n; ++p) any_limits = any_limits || (limits && limits[p]);
  ks::SweepItemArgs IA{};
  IA.ne = ne; IA.nr = nr; IA.T = T; IA.nc = nc; IA.nk = nk; IA.it_words = base->it_words; IA.req_words = base->req_words;
  IA.hot_words = (uint32_t)lay.c_hot_words(); IA.cold_words = (uint32_t)lay.cold_words();
  IA.bounds = bounds ? 1u : 0u; IA.has_topology = base->has_topology ? 1u : 0u; IA.pv_on = P.pv_on ? 1u : 0u; IA.hp_on = P.hp_on ? 1u : 0u;
  if (base->has_topology) {
    const ks::TopoView& Tv = P.topo;
    IA.G = (uint32_t)Tv.n_groups; IA.dom_words = (uint32_t)Tv.dom_words; IA.hg = (uint32_t)std::max(1, Tv.n_host_groups); IA.n_alias = (uint32_t)Tv.n_alias; IA.n_key_slots = (uint32_t)std::max(1, Tv.n_key_slots);
  }
  // every probe's descriptor and the bytes of its own block (the device lays the same block out with the same function)
  std::vector<size_t> own(n), hlb(n);
  for (uint32_t p = 0; p < n; ++p) {
    ks::SweepItemDesc& d = descs[p];
    d = ks::SweepItemDesc{};
    d.pod_b = pod_off[p]; d.m = pod_off[p + 1] - pod_off[p]; d.node_b = node_off[p]; d.n_removed = node_off[p + 1] - node_off[p];
    d.cb = claim_base[p]; d.mc = claim_base[p + 1] - claim_base[p];
    if (P.pv_on && ne) {
      size_t entries = 0;
      for (uint32_t i = pod_off[p]; i < pod_off[p + 1]; ++i) entries += base->h_pod_pv_first[pods[i] + 1] - base->h_pod_pv_first[pods[i]];
      d.pv_entries = (uint32_t)entries;
    }
    own[p] = ks::sweep_item_fill(nullptr, IA, d, p, &hlb[p]);
    if (cancel_h) cancel_h[p] = cancel[p];
  }
  auto layout = [&]() {
    off = 0;
    d_items = (ks::Workspace*)take((size_t)n * sizeof(ks::Workspace));
    d_descs = (ks::SweepItemDesc*)take((size_t)n * sizeof(ks::SweepItemDesc));
    d_cancel_each = any_cancel ? (volatile int**)take((size_t)n * 8) : nullptr;
    d_sorted = (uint32_t*)take((size_t)total_pods * 4 + 4);
    d_removed = (uint32_t*)take((size_t)total_nodes * 4 + 4);
    d_limits = any_limits ? (int64_t*)take((size_t)n * T * (nr + 1) * 8) : nullptr;
    d_order = (uint32_t*)take((size_t)n * 4 + 4);
    zero_from = off;                                       // everything from here on starts as zeroes
    d_next = (uint32_t*)take(64);
    d_slot = (uint32_t*)take((size_t)total_pods * 4 + 4); d_err = (uint8_t*)take(total_pods + 4); d_diag = (uint8_t*)take(total_pods + 4);
    d_last = (uint32_t*)take((size_t)total_pods * 4 + 4); d_queue = (uint32_t*)take(((size_t)total_pods + n) * 4 + 4);
    d_hot = (uint64_t*)take((size_t)total_mc * lay.c_hot_words() * 8); d_cold = (uint64_t*)take((size_t)total_mc * lay.cold_words() * 8);
    d_resv = (uint64_t*)take((size_t)total_mc * 8); d_chp = P.hp_on ? (uint64_t*)take((size_t)total_mc * 8) : nullptr;
    d_okey = (uint32_t*)take((size_t)total_mc * 4); d_oord = (uint32_t*)take((size_t)total_mc * 4); d_opos = (uint32_t*)take((size_t)total_mc * 4);
    d_nclaims = (int*)take((size_t)n * 4); d_status = (int*)take((size_t)n * 4); d_ctr = (ks::Counters*)take((size_t)n * sizeof(ks::Counters));
    for (uint32_t p = 0; p < n; ++p) { descs[p].off = off; off += own[p]; }
    zero_to = off;
    d_assign = (int32_t*)take((size_t)total_pods * 4 + 4);   // starts as -1; so do the hostname-group threshold bitmaps (all claims below every threshold)
    for (uint32_t p = 0; p < n; ++p) { descs[p].hl_off = off; off += hlb[p]; }
    ones_to = off;
  };
#ifdef KSOLVE_TEST_HOOKS
  auto t_b = tnow();
#endif
  layout();
  const size_t total = off;
  base->sweep_last_total = total;
  base->sweep_arena_refused = false;
  arena = sweep_buffer(base, base->sweep_arena, base->sweep_arena_bytes, total, true);
  if (!arena) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "device allocation failed (sweep arena)" : base->error);
  layout();
  std::vector<int64_t> lim;
  if (any_limits) {
    lim.resize((size_t)n * T * (nr + 1));
    std::vector<int64_t> base_lim((size_t)T * (nr + 1));
    be_d2h(base, base_lim.data(), P.tmpl_limits, base_lim.size() * 8);
    be_sync(base);
    for (uint32_t p = 0; p < n; ++p) memcpy(lim.data() + (size_t)p * T * (nr + 1), (limits[p] ? limits[p] : base_lim.data()), (size_t)T * (nr + 1) * 8);
  }
  IA.arena = arena; IA.desc = d_descs; IA.items = d_items; IA.cancel_each = (volatile int* const*)d_cancel_each;
  IA.d_sorted = d_sorted; IA.d_removed = d_removed; IA.d_slot = d_slot; IA.d_last = d_last; IA.d_queue = d_queue; IA.d_okey = d_okey; IA.d_oord = d_oord; IA.d_opos = d_opos;
  IA.d_limits = d_limits; IA.d_assign = d_assign; IA.d_err = d_err; IA.d_diag = d_diag;
  IA.d_hot = d_hot; IA.d_cold = d_cold; IA.d_resv = d_resv; IA.d_chp = d_chp; IA.d_nclaims = d_nclaims; IA.d_status = d_status; IA.d_ctr = d_ctr;
  IA.n_mask0 = base->ws.n_mask0; IA.n_defined0 = base->ws.n_defined0; IA.n_complement0 = base->ws.n_complement0; IA.n_remaining0 = base->ws.n_remaining0;
  IA.cancel = base->d_cancel; IA.max_steps = base->opts.max_steps; IA.min_values_best_effort = base->opts.min_values_best_effort ? 1 : 0; IA.order_cap = lp.order_cap;
  ks::Workspace item0 = ks::Workspace{};   // probe 0's record as the device builds it: the finalize kernel borrows its per-template arrays
  if (n) ks::sweep_item_fill(&item0, IA, descs[0], 0, nullptr);
#ifdef KSOLVE_TEST_HOOKS
  auto t_c = tnow();
#endif
  be_h2d(base, d_descs, descs, (size_t)n * sizeof(ks::SweepItemDesc));
  if (d_cancel_each) be_h2d(base, d_cancel_each, (const void*)cancel_h, (size_t)n * 8);
  be_launch_sweep_items(base, (int)n, IA);
  if (total_pods) be_h2d(base, d_sorted, sorted, (size_t)total_pods * 4);
  if (total_nodes) be_h2d(base, d_removed, removed, (size_t)total_nodes * 4);
  if (d_limits) be_h2d(base, d_limits, lim.data(), lim.size() * 8);
  {
    // The order the compact launch hands its probes out in (one atomic counter, ksolve_pack_sweep4): most displaced pods first. A
    // probe is a serial chain whose length goes with its pods; 2048 wavefronts take ~5 probes each of a 10k-probe sweep, and with
    // a fixed share per wavefront the launch lasted as long as its unluckiest wavefront (2.9 ms against 1.9 ms of mean work).
    // Largest first + whoever is free takes the next one ends the launch within the last, smallest probes of the mean.
    std::vector<uint32_t> start((size_t)max_m + 2, 0);
    for (uint32_t p = 0; p < n; ++p) start[(size_t)max_m - (pod_off[p + 1] - pod_off[p]) + 1]++;
    for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
    for (uint32_t p = 0; p < n; ++p) ord[start[(size_t)max_m - (pod_off[p + 1] - pod_off[p])]++] = p;
    be_h2d(base, d_order, ord, (size_t)n * 4);
  }
  be_fill(base, arena + zero_from, 0, zero_to - zero_from);
  be_fill(base, arena + zero_to, 0xFF, ones_to - zero_to);
  be_h2d(base, base->d_pv, &P, sizeof(P));
  be_sync(base);
#ifdef KSOLVE_TEST_HOOKS
  if (trace) {
    auto us_ = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    fprintf(stderr, "sweep_run host: validate+sort %.0f us, layout+items %.0f us, h2d+fill %.0f us, arena %zu bytes, sizeof(Workspace) %zu, LDS %d bytes per workgroup of %d wavefront(s)\n", us_(t_a, t_b), us_(t_b, t_c), us_(t_c, tnow()), total, sizeof(ks::Workspace), lp.total_bytes, lp.waves ? lp.waves : 1);
  }
#endif
  be_toc(base, T_UPLOAD);
  upload_range.open = false;
  if (!be_ok(base)) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "sweep upload failed" : base->error);

  // ---- the launch: block b = the general engine on probe b ----
  be_launch_pack_sweep(base, base->d_pv, d_items, (int)n, lp, d_order, d_next);
  std::vector<int> n_claims(n, 0), status(n, 0);
  be_d2h(base, n_claims.data(), d_nclaims, (size_t)n * 4);
  be_d2h(base, status.data(), d_status, (size_t)n * 4);
  std::vector<uint32_t> oord(total_mc);
  if (total_mc) be_d2h(base, oord.data(), d_oord, (size_t)total_mc * 4);
  be_sync(base);
  if (!be_ok(base)) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "sweep launch failed" : base->error);

  // ---- finalize over the claims of every probe, in the order the reference's slices end in ----
  be_tic(base, T_FINALIZE);
  im->claim_off.assign(n + 1, 0);
  im->status.assign(n, KSOLVE_OK);
  std::vector<uint32_t> slot_of;
  for (uint32_t p = 0; p < n; ++p) {
    if (status[p] == 1) { im->status[p] = KSOLVE_ERR_CAPACITY; n_claims[p] = 0; }
    else if (status[p] == 2) im->status[p] = KSOLVE_ERR_CANCELLED;
    for (int i = 0; i < n_claims[p]; ++i) slot_of.push_back(claim_base[p] + oord[claim_base[p] + i]);
    im->claim_off[p + 1] = (uint32_t)slot_of.size();
  }
  const uint32_t C = (uint32_t)slot_of.size();
  const uint32_t n_its = base->n_its;
  const bool trunc = base->opts.truncate_instance_types != 0;
  std::vector<uint64_t> hot((size_t)C * lay.c_hot_words()), cold((size_t)C * lay.cold_words()), reserved(C);
  std::vector<double> cheapest(C);
  std::vector<int64_t> daemon_req((size_t)C * nr);
  std::vector<int32_t> t_idx; std::vector<uint32_t> t_cnt; std::vector<uint8_t> t_fail;
  if (C) {
    size_t fo = 0;
    char* fb = nullptr;
    auto ftake = [&](size_t bytes) { void* q = fb ? (void*)(fb + fo) : nullptr; fo += (bytes + 63) & ~(size_t)63; return q; };
    uint32_t* f_slot; double* f_cheap; int64_t* f_dreq; int32_t* f_idx = nullptr; double* f_price = nullptr; uint32_t* f_cnt = nullptr; uint8_t* f_fail = nullptr;
    uint64_t *g_hot, *g_cold, *g_resv;
    auto flay = [&]() {
      fo = 0;
      f_slot = (uint32_t*)ftake((size_t)C * 4); f_cheap = (double*)ftake((size_t)C * 8); f_dreq = (int64_t*)ftake((size_t)C * nr * 8);
      if (trunc) { f_idx = (int32_t*)ftake((size_t)C * n_its * 4); f_price = (double*)ftake((size_t)C * n_its * 8); f_cnt = (uint32_t*)ftake((size_t)C * 4); f_fail = (uint8_t*)ftake(C); }
      g_hot = (uint64_t*)ftake(hot.size() * 8); g_cold = (uint64_t*)ftake(cold.size() * 8); g_resv = (uint64_t*)ftake((size_t)C * 8);
    };
    flay();
    fb = sweep_buffer(base, base->sweep_fin, base->sweep_fin_bytes, fo);
    if (!fb) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "device allocation failed (sweep finalize)" : base->error);
    flay();
    be_h2d(base, f_slot, slot_of.data(), (size_t)C * 4);
    ks::FinalizeArgs F{P.dict, (int)n_its, (int)base->it_words, P.n_zones, P.n_cts, P.it_off_avail, P.it_off_price, d_hot, d_cold, lay, f_cheap,
                       P.dg_first, P.dg_ov, P.dg_its, P.dg_nonempty, item0.t_its, f_dreq,
                       P.reserved_on ? d_resv : nullptr, P.it_resv_first, P.resv_zone, P.resv_id, P.resv_price,
                       0, 0, P.it_reqs, nullptr, nullptr, nullptr, nullptr, f_slot};
    if (trunc) {
      F.truncate_n = (int)base->opts.truncate_instance_types; F.best_effort = base->opts.min_values_best_effort ? 1 : 0;
      F.sort_idx = f_idx; F.sort_price = f_price; F.ordered_count = f_cnt; F.trunc_failed = f_fail;
    }
    be_launch_finalize(base, (int)C, F);
    ks::ClaimGatherArgs G{lay, f_slot, d_hot, d_cold, d_resv, g_hot, g_cold, g_resv};
    be_launch_claim_gather(base, (int)C, G);
    be_toc(base, T_FINALIZE);
    be_tic(base, T_DOWNLOAD);
    be_d2h(base, hot.data(), g_hot, hot.size() * 8); be_d2h(base, cold.data(), g_cold, cold.size() * 8); be_d2h(base, reserved.data(), g_resv, (size_t)C * 8);
    be_d2h(base, cheapest.data(), f_cheap, (size_t)C * 8); be_d2h(base, daemon_req.data(), f_dreq, daemon_req.size() * 8);
    if (trunc) {
      t_idx.resize((size_t)C * n_its); t_cnt.resize(C); t_fail.resize(C);
      be_d2h(base, t_idx.data(), f_idx, t_idx.size() * 4); be_d2h(base, t_cnt.data(), f_cnt, (size_t)C * 4); be_d2h(base, t_fail.data(), f_fail, C);
    }
  } else { be_toc(base, T_FINALIZE); be_tic(base, T_DOWNLOAD); }
  std::vector<int32_t> assign(total_pods); std::vector<uint32_t> slot(total_pods); std::vector<uint8_t> err(total_pods), diag(total_pods);
  if (total_pods) {
    be_d2h(base, assign.data(), d_assign, (size_t)total_pods * 4); be_d2h(base, slot.data(), d_slot, (size_t)total_pods * 4);
    be_d2h(base, err.data(), d_err, total_pods); be_d2h(base, diag.data(), d_diag, total_pods);
  }
  im->counters.resize(n);
  be_d2h(base, im->counters.data(), d_ctr, (size_t)n * sizeof(ks::Counters));
  be_sync(base);
  be_toc(base, T_DOWNLOAD);
  if (!be_ok(base)) return fail(base, KSOLVE_ERR_DEVICE, base->error.empty() ? "sweep download failed" : base->error);

  // ---- rows of ksolve_claims + per-pod results in the caller's pod order ----
  ClaimCols cc; cc.resize(C, base);
  for (uint32_t c = 0; c < C; ++c)
    decode_claim(base, P, c, hot.data() + (size_t)c * lay.c_hot_words(), cold.data() + (size_t)c * lay.cold_words(), reserved[c], daemon_req.data() + (size_t)c * nr, cc);
  ResultsImpl& R = im->claims;
  R.tmpl = std::move(cc.tmpl); R.npods = std::move(cc.npods); R.its = std::move(cc.its); R.mask = std::move(cc.mask);
  R.defined = std::move(cc.defined); R.complement = std::move(cc.complement); R.has_gte = std::move(cc.has_gte); R.has_lte = std::move(cc.has_lte);
  R.gte = std::move(cc.gte); R.lte = std::move(cc.lte); R.minv = std::move(cc.minv); R.requests = std::move(cc.requests);
  R.host_seq = std::move(cc.host_seq); R.relaxed = std::move(cc.relaxed); R.cheapest = std::move(cheapest); R.reserved = std::move(reserved);
  R.t_idx = std::move(t_idx); R.t_cnt = std::move(t_cnt); R.t_fail = std::move(t_fail);
  im->assign.assign(total_pods, -1); im->slot.assign(total_pods, 0); im->err.assign(total_pods, 0); im->diag.assign(total_pods, 0);
  im->ref.assign(n, 0);
  std::vector<uint32_t> newidx;
  for (uint32_t p = 0; p < n; ++p) {
    const uint32_t b = pod_off[p], m = pod_off[p + 1] - b, cb = claim_base[p];
    const uint32_t Cp = im->claim_off[p + 1] - im->claim_off[p];
    newidx.assign(claim_base[p + 1] - cb, 0);
    for (uint32_t i = 0; i < Cp; ++i) newidx[oord[cb + i]] = i;
    for (uint32_t i = 0; i < m; ++i) {
      const uint32_t dst = b + perm[b + i];   // sorted position i is the caller's pod perm[i]
      int32_t a = assign[b + i];
      if (im->status[p] == KSOLVE_ERR_CAPACITY) a = -1;
      else if (a >= 0) a = (int32_t)newidx[a];
      im->assign[dst] = a; im->slot[dst] = slot[b + i]; im->err[dst] = err[b + i]; im->diag[dst] = diag[b + i];
    }
    im->ref[p] = im->counters[p].ref_bin_evaluations;
  }
  if (us) { us[0] = base->timers.ms[T_UPLOAD] * 1e3; us[1] = base->timers.ms[T_PACK] * 1e3; us[2] = base->timers.ms[T_FINALIZE] * 1e3; us[3] = base->timers.ms[T_DOWNLOAD] * 1e3; }
  return KSOLVE_OK;
}

// Sweeps whose workspaces would not fit a sensible arena (a cluster with hostname topology groups needs a per-node counter table
// per probe) run as several launches; the results are those of one.
static void sweep_append(SweepImpl* im, const SweepImpl& part, uint32_t m);
// Bytes of arena probe p of a sweep needs: sweep_run's layout() written out per probe (its share of the per-pod and per-claim
// regions + its own workspace), every region rounded up as take() rounds it.
static size_t sweep_probe_bytes(const ksolve_handle* base, uint32_t m, size_t pv_entries, bool with_limits) {
  const ks::ProblemView& P = base->pv;
  const ks::RecLayout lay = P.lay;
  const size_t ne = base->n_nodes, nr = base->n_res, T = base->n_templates, nc = std::max(1u, base->n_classes), nk = base->n_keys;
  size_t mc = std::max(1u, m);
  if (base->opts.max_claims && base->opts.max_claims < mc) mc = base->opts.max_claims;
  const size_t cw = (mc + 63) / 64;
  size_t oc = 64;
  while (oc < 2 * std::min<size_t>(std::max<size_t>(1, m), std::max<size_t>(1, ne))) oc <<= 1;
  auto r = [](size_t b) { return (b + 63) & ~(size_t)63; };
  size_t b = sizeof(ks::Workspace) + sizeof(ks::SweepItemDesc) + 8 + 8 + 4 + sizeof(ks::Counters);   // + its descriptor, its cancel flag's address, nclaims, status, its place in the hand-out order
  b += (size_t)m * (4 + 4 + 1 + 1 + 4 + 4 + 4) + 4;                                     // sorted, slot, err, diag, lastLen, queue, assign
  b += mc * ((size_t)lay.c_hot_words() * 8 + (size_t)lay.cold_words() * 8 + 8 + (P.hp_on ? 8 : 0) + 12);
  if (with_limits) b += T * (nr + 1) * 8;
  b += r((mc * nr + 64) * 8) + r(T * base->it_words * 8) + r(T * (nr + 1) * 8) + r(nc * cw * 8);
  if (ne) {
    const bool bounds = base->ws.n_hg != nullptr;
    b += 2 * r(oc * 4) + r((size_t)base->req_words * oc * 8) + 2 * r(oc * 4) + r(nr * oc * 8) + r(oc * 4) + (P.hp_on ? r(oc * 8) : 0);
    if (bounds) b += 2 * r(oc * 4) + 2 * r(nk * oc * 8);
    if (base->has_topology) {
      const ks::TopoView& Tv = P.topo;
      const size_t G = (size_t)Tv.n_groups, dv = (size_t)Tv.dom_words * 64, hg = (size_t)std::max(1, Tv.n_host_groups), ks_ = (size_t)std::max(1, Tv.n_key_slots);
      b += r(G * Tv.dom_words * 8) + 2 * r(G * dv * 4) + r(hg * oc * 4) + r(hg * mc * 4) + r(G * 4) + (Tv.n_alias ? r((size_t)Tv.n_alias * 4) : 0);
      b += r(ks_ * mc * 8) + r(ks_ * 64 * cw * 8) + r(hg * 2 * cw * 8);
    }
    if (P.pv_on) b += r(std::max<size_t>(1, pv_entries) * 8);
  }
  return b + b / 64 + 2048;   // + the rounding of the ~20 shared regions (each up to 63 bytes, paid in full by a launch of one probe) and 1.5% of slack; the test builds fail a sweep whose arena outgrows this figure
}

// Sweeps whose workspaces would not fit the arena budget run as several launches; the results are those of one. The chunks are
// cut by the probes' MEASURED sizes (a multi-node prefix of 2000 pods needs ~1 MB, a single-node probe ~60 KB, a cluster with
// hostname topology groups a per-node counter table per probe); a chunk whose arena the device still refuses is halved and retried.
static ksolve_status sweep_run_chunked(ksolve_handle* base, uint32_t n, const uint32_t* node_off, const uint32_t* nodes, const uint32_t* pod_off, const uint32_t* pods,
                                       const int64_t* const* limits, int* const* cancel, SweepImpl* im, double* us) {
  for (uint32_t p = 0; p < n; ++p)
    if (node_off[p + 1] < node_off[p] || pod_off[p + 1] < pod_off[p]) return fail(base, KSOLVE_ERR_INVALID, "sweep descriptor offsets must not decrease");
  { ksolve_status st = sweep_prepare_base(base); if (st != KSOLVE_OK) return st; }   // the class count enters the sizes below
  size_t budget = (size_t)4 << 30;
  if (const char* b = getenv("KSOLVE_SWEEP_ARENA_MB")) budget = (size_t)std::max(1, atoi(b)) << 20;   // the arena budget of a launch (tests lower it to force several)
  bool any_limits = false;
  for (uint32_t p = 0; p < n; ++p) any_limits = any_limits || (limits && limits[p]);
  std::vector<size_t> need(n);
  size_t total_need = 0;
  for (uint32_t p = 0; p < n; ++p) {
    size_t entries = 0;
    if (base->pv.pv_on) for (uint32_t i = pod_off[p]; i < pod_off[p + 1]; ++i) if (pods[i] < base->n_pods) entries += base->h_pod_pv_first[pods[i] + 1] - base->h_pod_pv_first[pods[i]];
    need[p] = sweep_probe_bytes(base, pod_off[p + 1] - pod_off[p], entries, any_limits);
    total_need += need[p];
  }
  if (total_need <= budget) {
    ksolve_status st = sweep_run(base, n, node_off, nodes, pod_off, pods, limits, cancel, im, us);
#ifdef KSOLVE_TEST_HOOKS
    if (st == KSOLVE_OK && base->sweep_last_total > total_need) return fail(base, KSOLVE_ERR_INVALID, "sweep_probe_bytes underestimates the arena: " + std::to_string(base->sweep_last_total) + " > " + std::to_string(total_need) + " for " + std::to_string(n) + " probes");
#endif
    if (st != KSOLVE_ERR_DEVICE || n < 2 || !base->sweep_arena_refused) return st;   // (only a refused arena is retried, not any device error)
    budget = total_need / 2;            // the device refused the arena: go on with half of it per launch
    base->error.clear();
    *im = SweepImpl();
  }
  double total_us[4] = {0, 0, 0, 0};
  im->claim_off.assign(1, 0);
  for (uint32_t lo = 0; lo < n;) {
    uint32_t m = 0;
    size_t acc = 0;
    while (lo + m < n && (m == 0 || acc + need[lo + m] <= budget)) { acc += need[lo + m]; ++m; }
    std::vector<uint32_t> no(m + 1), po(m + 1);
    for (uint32_t i = 0; i <= m; ++i) { no[i] = node_off[lo + i] - node_off[lo]; po[i] = pod_off[lo + i] - pod_off[lo]; }
    SweepImpl part;
    double u[4] = {0, 0, 0, 0};
    ksolve_status st = sweep_run(base, m, no.data(), nodes + node_off[lo], po.data(), pods + pod_off[lo], limits ? limits + lo : nullptr, cancel ? cancel + lo : nullptr, &part, u);
    if (st == KSOLVE_ERR_DEVICE && m > 1 && base->sweep_arena_refused) { budget = std::max<size_t>(acc / 2, 1); base->error.clear(); continue; }   // smaller launches from here on
    if (st != KSOLVE_OK) return st;
#ifdef KSOLVE_TEST_HOOKS   // the test builds hold the estimate to the layout it restates
    if (base->sweep_last_total > acc) return fail(base, KSOLVE_ERR_INVALID, "sweep_probe_bytes underestimates the arena");
#endif
    lo += m;
    for (int k = 0; k < 4; ++k) total_us[k] += u[k];
    sweep_append(im, part, m);
  }
  if (us) for (int k = 0; k < 4; ++k) us[k] = total_us[k];
  return KSOLVE_OK;
}

static void fill_claims_view(const ksolve_handle* h, const ResultsImpl& R, size_t first, uint32_t count, ksolve_claims& cl) {
  cl.n_claims = count; cl.it_words = h->it_words; cl.req_words = h->req_words; cl.n_keys = h->n_keys; cl.n_res = h->n_res;
  cl.template_idx = R.tmpl.data() + first; cl.pod_count = R.npods.data() + first; cl.it_mask = R.its.data() + first * h->it_words; cl.requests = R.requests.data() + first * h->n_res;
  cl.req_mask = R.mask.data() + first * h->req_words; cl.req_defined = R.defined.data() + first; cl.req_complement = R.complement.data() + first;
  cl.req_has_gte = R.has_gte.data() + first; cl.req_has_lte = R.has_lte.data() + first; cl.req_gte = R.gte.data() + first * h->n_keys; cl.req_lte = R.lte.data() + first * h->n_keys;
  cl.req_min_values = R.minv.data() + first * h->n_keys; cl.min_values_relaxed = R.relaxed.data() + first; cl.cheapest_price = R.cheapest.data() + first;
  cl.hostname_seq = R.host_seq.data() + first; cl.reserved_mask = R.reserved.data() + first;
  cl.n_instance_types = h->n_its;
  if (!R.t_cnt.empty()) { cl.ordered_instance_types = R.t_idx.data() + first * h->n_its; cl.ordered_count = R.t_cnt.data() + first; cl.truncation_failed = R.t_fail.data() + first; }
}

// ksolve_sweep
static ksolve_status sweep(ksolve_handle* base, const ksolve_sweep_desc* d, ksolve_sweep_results* out) {
  memset(out, 0, sizeof(*out));
  if (!base || base->base || !d) return base ? fail(base, KSOLVE_ERR_INVALID, "sweep of a null handle / of a probe") : KSOLVE_ERR_INVALID;
  if (base->has_topology && !base->resident) return fail(base, KSOLVE_ERR_UNSUPPORTED, "sweeps of a problem with topology groups need a resident-cluster base (ksolve_problem_desc.pod_node): its counts include the candidates' pods");
  if (d->n_probes && (!d->node_off || !d->pod_off || (d->pod_off[d->n_probes] && !d->pods) || (d->node_off[d->n_probes] && !d->nodes)))
    return fail(base, KSOLVE_ERR_INVALID, "sweep descriptor arrays missing");
  const uint32_t n = d->n_probes;
  const size_t lsz = (size_t)base->n_templates * (base->n_res + 1);
  std::vector<const int64_t*> lims(n, nullptr);
  if (d->tmpl_limits) for (uint32_t p = 0; p < n; ++p) lims[p] = d->tmpl_limits + (size_t)p * lsz;
  SweepImpl* im = new SweepImpl();
  double us[4] = {0, 0, 0, 0};
  const uint32_t zero_off[1] = {0};
  be_fill(base, base->d_cancel, 0, 4);   // a fresh context, as in solve_prepare
  be_sync(base);
  ksolve_status st = n ? sweep_run_chunked(base, n, d->node_off, d->nodes, d->pod_off, d->pods, d->tmpl_limits ? lims.data() : nullptr, nullptr, im, us) : KSOLVE_OK;
  if (!n) { im->claim_off.assign(1, 0); (void)zero_off; }
  if (st != KSOLVE_OK) { delete im; return st; }
  out->n_probes = n;
  out->status = im->status.data();
  out->pod_assignment = im->assign.data(); out->pod_error = im->err.data(); out->pod_error_diag = im->diag.data(); out->pod_slot = im->slot.data();
  out->claim_off = im->claim_off.data();
  fill_claims_view(base, im->claims, 0, im->claim_off[n], out->claims);
  out->ref_bin_evaluations = im->ref.data();
  out->us_upload = us[0]; out->us_pack = us[1]; out->us_finalize = us[2]; out->us_download = us[3];
  out->n_classes = base->n_classes; out->us_node_dead0 = base->dead0_us; out->it_words = base->it_words; out->n_nodes = base->n_nodes;
  for (auto& c : im->counters) { out->total_bin_evaluations += c.bin_evaluations; out->total_node_evaluations += c.node_evaluations; out->total_node_block_steps += c.node_block_steps; }
  out->impl = im;
  return KSOLVE_OK;
}

// Appends the results of the probes of `part` behind those already in `im` (chunks of one sweep, shares of several devices).
static void sweep_append(SweepImpl* im, const SweepImpl& part, uint32_t m) {
  auto app = [](auto& dst, const auto& src) { dst.insert(dst.end(), src.begin(), src.end()); };
  app(im->status, part.status); app(im->assign, part.assign); app(im->err, part.err); app(im->diag, part.diag); app(im->slot, part.slot);
  app(im->ref, part.ref); app(im->counters, part.counters);
  const uint32_t c0 = im->claim_off.back();
  for (uint32_t i = 1; i <= m; ++i) im->claim_off.push_back(c0 + part.claim_off[i]);
  ResultsImpl& R = im->claims; const ResultsImpl& Q = part.claims;
  app(R.tmpl, Q.tmpl); app(R.npods, Q.npods); app(R.its, Q.its); app(R.mask, Q.mask); app(R.defined, Q.defined); app(R.complement, Q.complement);
  app(R.has_gte, Q.has_gte); app(R.has_lte, Q.has_lte); app(R.gte, Q.gte); app(R.lte, Q.lte); app(R.minv, Q.minv); app(R.requests, Q.requests);
  app(R.host_seq, Q.host_seq); app(R.relaxed, Q.relaxed); app(R.cheapest, Q.cheapest); app(R.reserved, Q.reserved);
  app(R.t_idx, Q.t_idx); app(R.t_cnt, Q.t_cnt); app(R.t_fail, Q.t_fail);
}

// ksolve_sweep_replicas: ONE sweep over several devices. `bases` are resident-cluster handles created from the SAME problem on
// different devices (ksolve_options.device; the cluster tables are replicated — 0.6 GB at 100k nodes / 2M pods against 288 GB of
// HBM per device — so that no probe needs anything from another device). The probes are cut into contiguous shares of about equal
// displaced-pod counts, every device runs its share on its own host thread (one or several launches), and the results come back
// in probe order as if one device had run them all: a one-process caller (the Go controller that owns the node's eight GPUs)
// needs no collective for configs[4]. SimulateScheduling per probe: disruption/helpers.go:53-155.
static ksolve_status sweep_replicas(ksolve_handle** bases, uint32_t nb, const ksolve_sweep_desc* d, ksolve_sweep_results* out) {
  if (nb == 1) return sweep(bases[0], d, out);
  memset(out, 0, sizeof(*out));
  ksolve_handle* b0 = bases[0];
  for (uint32_t g = 0; g < nb; ++g) {
    ksolve_handle* b = bases[g];
    if (!b || b->base) return fail(b0, KSOLVE_ERR_INVALID, "sweep over a null handle / a probe");
    if (b->n_pods != b0->n_pods || b->n_nodes != b0->n_nodes || b->n_templates != b0->n_templates || b->n_res != b0->n_res || b->n_its != b0->n_its || b->req_words != b0->req_words)
      return fail(b0, KSOLVE_ERR_INVALID, "the handles of a replicated sweep must be replicas of one problem");
    if (b->has_topology && !b->resident) return fail(b0, KSOLVE_ERR_UNSUPPORTED, "sweeps of a problem with topology groups need a resident-cluster base");
  }
  if (!d || (d->n_probes && (!d->node_off || !d->pod_off || (d->pod_off[d->n_probes] && !d->pods) || (d->node_off[d->n_probes] && !d->nodes))))
    return fail(b0, KSOLVE_ERR_INVALID, "sweep descriptor arrays missing");
  const uint32_t n = d->n_probes;
  for (uint32_t p = 0; p < n; ++p)
    if (d->node_off[p + 1] < d->node_off[p] || d->pod_off[p + 1] < d->pod_off[p]) return fail(b0, KSOLVE_ERR_INVALID, "sweep descriptor offsets must not decrease");
  // contiguous shares of about equal work (displaced pods + one per probe)
  std::vector<uint32_t> cut(nb + 1, n);
  cut[0] = 0;
  {
    const uint64_t total = (uint64_t)d->pod_off[n] + n;
    uint32_t p = 0;
    for (uint32_t g = 1; g < nb; ++g) {
      const uint64_t want = total * g / nb;
      while (p < n && (uint64_t)d->pod_off[p] + p < want) ++p;
      cut[g] = p;
    }
  }
  const size_t lsz = (size_t)b0->n_templates * (b0->n_res + 1);
  std::vector<SweepImpl> parts(nb);
  std::vector<ksolve_status> rc(nb, KSOLVE_OK);
  std::vector<std::array<double, 4>> uss(nb);
  std::vector<std::thread> pool;
  for (uint32_t g = 0; g < nb; ++g) pool.emplace_back([&, g]() {
    const uint32_t lo = cut[g], m = cut[g + 1] - lo;
    uss[g] = {0, 0, 0, 0};
    if (!m) { parts[g].claim_off.assign(1, 0); return; }
    ksolve_handle* b = bases[g];
    be_thread_init(b);
    std::vector<uint32_t> no(m + 1), po(m + 1);
    for (uint32_t i = 0; i <= m; ++i) { no[i] = d->node_off[lo + i] - d->node_off[lo]; po[i] = d->pod_off[lo + i] - d->pod_off[lo]; }
    std::vector<const int64_t*> lims(m, nullptr);
    if (d->tmpl_limits) for (uint32_t i = 0; i < m; ++i) lims[i] = d->tmpl_limits + (size_t)(lo + i) * lsz;
    be_fill(b, b->d_cancel, 0, 4);
    be_sync(b);
    rc[g] = sweep_run_chunked(b, m, no.data(), d->nodes + d->node_off[lo], po.data(), d->pods + d->pod_off[lo], d->tmpl_limits ? lims.data() : nullptr, nullptr, &parts[g], uss[g].data());
  });
  for (auto& th : pool) th.join();
  for (uint32_t g = 0; g < nb; ++g) if (rc[g] != KSOLVE_OK) return fail(b0, rc[g], "device " + std::to_string(g) + " of the sweep: " + bases[g]->error);
  SweepImpl* im = new SweepImpl();
  im->claim_off.assign(1, 0);
  for (uint32_t g = 0; g < nb; ++g) if (cut[g + 1] > cut[g]) sweep_append(im, parts[g], cut[g + 1] - cut[g]);
  out->n_probes = n;
  out->status = im->status.data();
  out->pod_assignment = im->assign.data(); out->pod_error = im->err.data(); out->pod_error_diag = im->diag.data(); out->pod_slot = im->slot.data();
  out->claim_off = im->claim_off.data();
  fill_claims_view(b0, im->claims, 0, im->claim_off[n], out->claims);
  out->ref_bin_evaluations = im->ref.data();
  for (uint32_t g = 0; g < nb; ++g) {   // the devices run side by side: the call takes as long as the slowest
    out->us_upload = std::max(out->us_upload, uss[g][0]); out->us_pack = std::max(out->us_pack, uss[g][1]);
    out->us_finalize = std::max(out->us_finalize, uss[g][2]); out->us_download = std::max(out->us_download, uss[g][3]);
  }
  out->n_classes = b0->n_classes; out->us_node_dead0 = b0->dead0_us; out->it_words = b0->it_words; out->n_nodes = b0->n_nodes;
  for (auto& c : im->counters) { out->total_bin_evaluations += c.bin_evaluations; out->total_node_evaluations += c.node_evaluations; out->total_node_block_steps += c.node_block_steps; }
  out->impl = im;
  return KSOLVE_OK;
}

// ksolve_packing_vector: per instance type, the NodeClaims that launch on it and their $/h. A claim launches on the type that
// gives its cheapest launch price — the cheapest available offering its requirements admit (OrderByPrice's key, types.go:336-355;
// ties: the lower instance-type index) — and is booked there with that price (cheapest_price of the Results).
static ksolve_status packing_vector(const ksolve_handle* h, const ksolve_claims& cl, double* count, double* cost) {
  const ksolve_handle* b = h->base ? h->base : h;
  const uint32_t n_its = b->n_its;
  for (uint32_t i = 0; i < n_its; ++i) { count[i] = 0; cost[i] = 0; }
  ks::Dict d = b->pv.dict;
  d.value_int = b->h_value_int.data(); d.value_is_int = b->h_value_is_int.data(); d.value_valid = nullptr;
  const int nk = (int)b->n_keys;
  for (uint32_t c = 0; c < cl.n_claims; ++c) {
    ks::ReqRef r;
    r.mask = cl.req_mask + (size_t)c * cl.req_words; r.defined = cl.req_defined[c]; r.complement = cl.req_complement[c];
    r.has_gte = cl.req_has_gte[c]; r.has_lte = cl.req_has_lte[c]; r.gte = cl.req_gte + (size_t)c * nk; r.lte = cl.req_lte + (size_t)c * nk; r.minv = nullptr;
    uint32_t zones = 0, cts = 0;
    if (d.key_zone >= 0 && ks::bit(r.defined, d.key_zone)) { for (int z = 0; z < b->h_n_zones; ++z) if (ks::req_has(d, r, d.key_zone, d.key_word_off[d.key_zone], z)) zones |= 1u << z; }
    else zones = (1u << b->h_n_zones) - 1;
    if (d.key_ct >= 0 && ks::bit(r.defined, d.key_ct)) { for (int t = 0; t < b->h_n_cts; ++t) if (ks::req_has(d, r, d.key_ct, d.key_word_off[d.key_ct], t)) cts |= 1u << t; }
    else cts = (1u << b->h_n_cts) - 1;
    uint64_t cells = 0;
    for (uint32_t zz = zones; zz; zz &= zz - 1) cells |= (uint64_t)cts << (__builtin_ctz(zz) * 4);
    double best = 1.7976931348623157e308;
    int best_it = -1;
    for (uint32_t it = 0; it < n_its; ++it) {
      if (!((cl.it_mask[(size_t)c * cl.it_words + it / 64] >> (it % 64)) & 1)) continue;
      for (uint64_t av = b->h_it_off_avail[it] & cells; av; av &= av - 1) {
        const double p = b->h_it_off_price[(size_t)it * 64 + __builtin_ctzll(av)];
        if (p < best) { best = p; best_it = (int)it; }
      }
    }
    if (best_it < 0) continue;   // nothing launchable (reserved-only claims keep their price in cheapest_price; not booked per type)
    count[best_it] += 1;
    cost[best_it] += cl.cheapest_price[c] < 1e300 ? cl.cheapest_price[c] : best;
  }
  return KSOLVE_OK;
}

static ksolve_status packing_vector_sum(ksolve_handle* const* hs, const ksolve_results* rs, uint32_t n, double* count, double* cost) {
  const uint32_t n_its = (hs[0]->base ? hs[0]->base : hs[0])->n_its;
  std::vector<double> c(n_its), d(n_its);
  for (uint32_t i = 0; i < n_its; ++i) { count[i] = 0; cost[i] = 0; }
  for (uint32_t p = 0; p < n; ++p) {
    if ((hs[p]->base ? hs[p]->base : hs[p])->n_its != n_its) return fail(hs[0], KSOLVE_ERR_INVALID, "packing vectors of different catalogue sizes");
    packing_vector(hs[p], rs[p].claims, c.data(), d.data());
    for (uint32_t i = 0; i < n_its; ++i) { count[i] += c[i]; cost[i] += d[i]; }
  }
  return KSOLVE_OK;
}

// ksolve_results of ONE probe handle from its slice of a sweep: the contract of ksolve_probe_create — the base problem's pod
// numbering, pods outside the probe unassigned
static ksolve_status probe_results(ksolve_handle* h, const SweepImpl& S, uint32_t p, uint32_t pod_base, ksolve_results* out) {
  memset(out, 0, sizeof(*out));
  ksolve_handle* base = h->base;
  if (S.status[p] == KSOLVE_ERR_CAPACITY) return fail(h, KSOLVE_ERR_CAPACITY, "more in-flight NodeClaims than a probe keeps resident (ksolve_options.max_claims / the LDS claim order)");
  ResultsImpl* im = new ResultsImpl();
  const uint32_t n_pods = base->n_pods;
  im->assign.assign(n_pods, -1); im->err.assign(n_pods, 0); im->diag.assign(n_pods, 0); im->slot.assign(n_pods, 0);
  for (size_t i = 0; i < h->pr_pods.size(); ++i) {
    const uint32_t g = h->pr_pods[i];
    im->assign[g] = S.assign[pod_base + i]; im->err[g] = S.err[pod_base + i]; im->diag[g] = S.diag[pod_base + i]; im->slot[g] = S.slot[pod_base + i];
  }
  const uint32_t c0 = S.claim_off[p], C = S.claim_off[p + 1] - c0;
  const ResultsImpl& R = S.claims;
  auto cut = [&](auto& dst, const auto& src, size_t width) { if (!src.empty()) dst.assign(src.begin() + (size_t)c0 * width, src.begin() + (size_t)(c0 + C) * width); };
  cut(im->tmpl, R.tmpl, 1); cut(im->npods, R.npods, 1); cut(im->its, R.its, h->it_words); cut(im->mask, R.mask, h->req_words);
  cut(im->defined, R.defined, 1); cut(im->complement, R.complement, 1); cut(im->has_gte, R.has_gte, 1); cut(im->has_lte, R.has_lte, 1);
  cut(im->gte, R.gte, h->n_keys); cut(im->lte, R.lte, h->n_keys); cut(im->minv, R.minv, h->n_keys); cut(im->requests, R.requests, h->n_res);
  cut(im->host_seq, R.host_seq, 1); cut(im->relaxed, R.relaxed, 1); cut(im->cheapest, R.cheapest, 1); cut(im->reserved, R.reserved, 1);
  cut(im->t_idx, R.t_idx, h->n_its); cut(im->t_cnt, R.t_cnt, 1); cut(im->t_fail, R.t_fail, 1);
  double cost = 0;
  for (uint32_t i = 0; i < C; ++i) if (im->cheapest[i] < 1e300) cost += im->cheapest[i];
  out->status = (ksolve_status)S.status[p];
  out->n_pods = n_pods;
  out->pod_assignment = im->assign.data(); out->pod_error = im->err.data(); out->pod_error_diag = im->diag.data(); out->pod_slot = im->slot.data();
  fill_claims_view(h, *im, 0, C, out->claims);
  const ks::Counters& ctr = S.counters[p];
  out->bin_evaluations = ctr.bin_evaluations; out->it_evaluations = ctr.it_evaluations; out->queue_pops = ctr.queue_pops;
  out->sorts = ctr.sorts; out->slow_sorts = ctr.slow_sorts; out->relaxations = ctr.relaxations;
  out->ref_bin_evaluations = ctr.ref_bin_evaluations;
  for (int i = 0; i < 24; ++i) out->phase_cycles[i] = ctr.cycles[i];
  out->us_upload = base->timers.ms[T_UPLOAD] * 1e3; out->us_pack = base->timers.ms[T_PACK] * 1e3;
  out->us_finalize = base->timers.ms[T_FINALIZE] * 1e3; out->us_download = base->timers.ms[T_DOWNLOAD] * 1e3;
  h->timers = base->timers;
  out->packing_cost = cost;
  out->engine_used = 1; out->engine_fallback_reason = 0;
  out->impl = im;
  return out->status;
}

// probe handles of one base, one launch
static void solve_probes(ksolve_handle** hs, uint32_t n, ksolve_results* outs, ksolve_status* st, bool fresh_context) {
  ksolve_handle* base = hs[0]->base;
  std::vector<uint32_t> node_off(n + 1, 0), pod_off(n + 1, 0), nodes, pods;
  std::vector<const int64_t*> lims(n, nullptr);
  std::vector<int*> cancel(n, nullptr);
  bool any_lim = false;
  for (uint32_t i = 0; i < n; ++i) {
    ksolve_handle* h = hs[i];
    nodes.insert(nodes.end(), h->pr_nodes.begin(), h->pr_nodes.end()); pods.insert(pods.end(), h->pr_pods.begin(), h->pr_pods.end());
    node_off[i + 1] = (uint32_t)nodes.size(); pod_off[i + 1] = (uint32_t)pods.size();
    if (!h->pr_limits.empty()) { lims[i] = h->pr_limits.data(); any_lim = true; }
    cancel[i] = h->d_cancel;
    if (fresh_context) {
      be_fill(h, h->d_cancel, 0, 4);
#ifdef KSOLVE_TEST_HOOKS
      if (const char* at = getenv("KSOLVE_TEST_CANCEL_AT")) { const int v = -atoi(at); if (v < 0) be_h2d(h, h->d_cancel, &v, 4); }
#endif
      be_sync(h);
    }
  }
  SweepImpl S;
  ksolve_status rc = sweep_run_chunked(base, n, node_off.data(), nodes.data(), pod_off.data(), pods.data(), any_lim ? lims.data() : nullptr, cancel.data(), &S, nullptr);
  for (uint32_t i = 0; i < n; ++i) {
    if (rc != KSOLVE_OK) { memset(&outs[i], 0, sizeof(outs[i])); st[i] = fail(hs[i], rc, base->error); outs[i].status = st[i]; continue; }
    st[i] = probe_results(hs[i], S, i, pod_off[i], &outs[i]);
    outs[i].status = st[i];
  }
}

// Memory plan of the cursor engine. 0: claim records (32 B with one row of class slots, 56 B with four) + order arrays (6 B) per
// claim in LDS, ~3,000 claims beside the caches; 1: the claim records and the slow sorts' snapshot in HBM (FastWork::c_rec / o_snap), only the
// order's keys and ids in LDS: ~27,000 claims; 2: the order arrays in HBM too (FastWork::o_key / o_ord / o_snap): 65,472 claims, the range of the 16-bit claim ids.
// rows: 1 while at most 64 pod classes are ever live at once in the queue (counted before the loop), ks::kFastRows otherwise.
static void fast_plan_set(ksolve_handle* h, int plan, int rows) {
  const bool wide = plan >= 1;
  auto align = [](int x) { return (x + 15) & ~15; };
  ks::FastPlan& fp = h->fw.plan;
  rows = rows <= 1 ? 1 : ks::kFastRows;
  fp.rows = rows;
  fp.helper = (plan == 0 && rows == 1 && h->opts.engine == 5) ? 1 : 0;   // the two-wavefront kernel: on request only (engine 5) — measured 5.7% SLOWER than one wavefront on the headline problem (profiles/round5/pass_i)
  const int rec_bytes = rows == 1 ? (int)sizeof(ks::FastRec<1>) : (int)sizeof(ks::FastRec<ks::kFastRows>);
  int off = 0;
  fp.off_ent = off; off = align(off + ks::kFastEnt * (int)sizeof(ks::FastEnt));
  fp.off_pool = off; off = align(off + ks::kFastPool * 16);
  fp.off_slot = off; off = align(off + rows * 64 * (int)sizeof(ks::FastSlot));
  fp.off_misc = off; off = align(off + (int)sizeof(ks::FastMisc));
  fp.off_hot = off; off = align(off + (int)sizeof(ks::FastHot));
  const int budget = 160 * 1024 - 512;
  const int per_claim = wide ? 4 : rec_bytes + 6;   // plan 1: the order's keys and ids; plan 0: the record, the order and the order's snapshot
  int cap = plan >= 2 ? 65472 : ((budget - off - 64) / per_claim) & ~63;
  if (cap > 65472) cap = 65472;
  if (!wide && h->opts.lds_claim_cap && (int)((h->opts.lds_claim_cap + 63) & ~63u) < cap) cap = (int)((h->opts.lds_claim_cap + 63) & ~63u);
#ifdef KSOLVE_TEST_HOOKS
  if (plan == 1) if (const char* e = getenv("KSOLVE_TEST_WIDE_CAP")) { const int c = (atoi(e) + 63) & ~63; if (c > 0 && c < cap) cap = c; }   // tests: a small plan 1, so that small problems reach plan 2
#endif
  if (cap > (int)((h->fast_mc + 63) & ~63u)) cap = (int)((h->fast_mc + 63) & ~63u);
  fp.cap = cap;
  fp.global_state = plan >= 2 ? 2 : wide ? 1 : 0;
  fp.off_state = off; if (!wide) off = align(off + cap * rec_bytes);
  const int lds_order = plan >= 2 ? 0 : cap;
  fp.off_key = off; off = align(off + lds_order * 2);
  fp.off_ord = off; off = align(off + lds_order * 2);
  fp.off_snap = off; if (!wide) off = align(off + lds_order * 2);   // (plans 1, 2: the snapshot is FastWork::o_snap)
  fp.total_bytes = off;
}

static void be_results_drop(ksolve_results* r) { if (r && r->impl) { delete (ResultsImpl*)r->impl; r->impl = nullptr; } }
static ksolve_status solve(ksolve_handle* h, ksolve_results* out, bool fresh_context = true) {
  memset(out, 0, sizeof(*out));
  if (h->resident && h->has_topology) return fail(h, KSOLVE_ERR_INVALID, "a resident cluster with topology groups is solved through probes (ksolve_sweep / ksolve_probe_create): its counts include the bound pod rows");
  if (h->base) { ksolve_status st = KSOLVE_OK; solve_probes(&h, 1, out, &st, fresh_context); return st; }
  ksolve_status st = solve_prepare(h, fresh_context);
  if (st != KSOLVE_OK) return st;
  if (h->opts.engine == 6 && !(h->tw.enabled && h->n_pods && h->n_classes))
    return fail(h, KSOLVE_ERR_UNSUPPORTED, "spread engine requested for a problem outside its shape (no topology groups / existing nodes / minValues / reservations / relaxation rows)");
  if (h->opts.engine >= 2 && h->opts.engine != 6 && !(h->fw.enabled && !h->pv.big && h->n_pods && h->n_classes))
    return fail(h, KSOLVE_ERR_UNSUPPORTED, "cursor engine requested for a problem outside its shape (topology / existing nodes / minValues / reservations / relaxation rows)");
  if (h->tw.enabled && h->n_pods && h->n_classes) {
    // the spread engine first; status 3 = "not my shape / stopped before any result": the general engine takes over
    alloc_run_order(h);
    be_tic(h, T_PACK);
    be_launch_pack_topo(h);
    be_toc(h, T_PACK);
    int status = 0, n_claims = 0;
    be_d2h(h, &status, h->ws.status_out, 4);
    be_d2h(h, &n_claims, h->ws.n_claims_out, 4);
    be_sync(h);
    if (be_ok(h) && status != 3 && status != 1) {
      h->engine_used = 3;
      h->fast_reason = 0; h->fast_attempts = 1;
      if (n_claims) be_launch_fast_records(h, n_claims);
      return solve_finish(h, out);
    }
    if (be_ok(h) && status == 3) {
      ks::Counters ctr{};
      be_d2h(h, &ctr, h->ws.counters, sizeof(ctr));
      be_sync(h);
      h->fast_reason = (uint32_t)ctr.cycles[20];
    } else if (be_ok(h)) h->fast_reason = 100;
    if (h->opts.engine == 6) return fail(h, KSOLVE_ERR_UNSUPPORTED, "spread engine declined the problem (reason " + std::to_string(h->fast_reason) + ")");
    if (h->fast_reason != 27 && h->fast_reason != 100) h->tw.enabled = 0;   // not its shape: later solves of this handle go straight to the general engine
    st = solve_prepare(h, false);
    if (st != KSOLVE_OK) return st;
  }
  if (h->fw.enabled && !h->pv.big && h->n_pods && h->n_classes) {
    // the cursor engine first; status 3 = "not my shape / stopped before any result": the general engine takes over — unless all
    // that stopped it was the number of claims its LDS plan holds (reason 26): then once more with the claims' state in HBM
    h->fast_attempts = 0;
    for (;;) {
      h->fast_attempts++;
      be_tic(h, T_PACK);
      be_launch_pack_fast(h);
      be_toc(h, T_PACK);
      int status = 0, n_claims = 0;
      be_d2h(h, &status, h->ws.status_out, 4);
      be_d2h(h, &n_claims, h->ws.n_claims_out, 4);
      be_sync(h);
      if (be_ok(h) && status != 3 && status != 1) {
        h->engine_used = 2;
        h->fast_reason = 0;   // (a first attempt that ran out of LDS claim slots is not a fallback to the general engine)
        if (n_claims) be_launch_fast_records(h, n_claims);
        return solve_finish(h, out);
      }
      unsigned long long ctr_pops = 0;
      if (be_ok(h) && status == 3) {
        ks::Counters ctr{};
        be_d2h(h, &ctr, h->ws.counters, sizeof(ctr));
        be_sync(h);
        h->fast_reason = (uint32_t)ctr.cycles[20];
        ctr_pops = ctr.queue_pops;
      } else if (be_ok(h)) h->fast_reason = 100;   // more claims than max_claims: the general engine reports it (or moves to BIG)
      if (be_ok(h) && h->fast_reason == 26 && h->fw.plan.global_state < 2 && !h->opts.lds_claim_cap) {
        // out of claim slots: the next plan (later solves of this handle start there). The attempt says how many pods its
        // claims took: when the whole queue, at that rate and a quarter more, would not fit plan 1 either, go to plan 2 at once.
        const int had = h->fw.plan.cap;
        int next = h->fw.plan.global_state + 1;
        if (next == 1) {
          fast_plan_set(h, 1, h->fw.plan.rows);
          const double placed = (double)(ctr_pops > 0 ? ctr_pops : 1);
          const bool holds_all = h->fw.plan.cap >= (int)((h->fast_mc + 63) & ~63u);   // plan 1 has a slot for every claim the problem may open
          if (!holds_all && (double)had * (double)h->n_pods / placed * 1.25 > (double)h->fw.plan.cap) next = 2;
        }
        fast_plan_set(h, next, h->fw.plan.rows);
        if (h->fw.plan.cap > had) {
          st = solve_prepare(h, false);
          if (st != KSOLVE_OK) return st;
          continue;
        }
      }
      break;
    }
    if (h->opts.engine >= 2) return fail(h, KSOLVE_ERR_UNSUPPORTED, "cursor engine declined the problem (reason " + std::to_string(h->fast_reason) + ")");
    h->fw.enabled = 0;   // later solves of this handle go straight to the general engine
    st = solve_prepare(h, false);
    if (st != KSOLVE_OK) return st;
  }
  be_tic(h, T_PACK);
  if (h->n_pods) be_launch_pack(h);
  be_toc(h, T_PACK);
  if (!h->pv.big && h->big_capable) {
    int status = 0;
    be_d2h(h, &status, h->ws.status_out, 4);
    be_sync(h);
    if (status == 1) {
      // more in-flight claims than the LDS-resident order holds: this problem runs on the BIG engine from now on
      h->pv.big = 1; h->pv.lite = 0; h->pv.lds = h->lds_big;
      alloc_run_order(h);
      st = solve_prepare(h, false);
      if (st != KSOLVE_OK) return st;
      be_tic(h, T_PACK);
      be_launch_pack(h);
      be_toc(h, T_PACK);
    }
  }
  return solve_finish(h, out);
}

// Many independent problems, one launch: block b of the pack kernel is the wavefront of problem b. This is how
// consolidation sweeps (helpers.go:53-155: one Solve() per candidate set) and NodePool components fill the chip.
static ksolve_status solve_batch_plain(ksolve_handle** hs, uint32_t n, ksolve_results* outs) {
  for (uint32_t i = 0; i < n; ++i) memset(&outs[i], 0, sizeof(outs[i]));
  std::vector<ksolve_status> st(n, KSOLVE_OK);
  // The prepass (classes, queue order) and the result download of different problems are independent: a few host
  // threads drive them on the problems' own streams so that the launches and copies of different problems overlap.
  const uint32_t n_threads = std::min<uint32_t>(n, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
  auto parallel = [&](auto&& fn) {
    if (n_threads <= 1) { for (uint32_t i = 0; i < n; ++i) fn(i); return; }
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; ++t) pool.emplace_back([&, t]() { be_thread_init(hs[0]); for (uint32_t i = t; i < n; i += n_threads) fn(i); });
    for (auto& th : pool) th.join();
  };
  parallel([&](uint32_t i) { st[i] = solve_prepare(hs[i]); outs[i].status = st[i]; if (st[i] == KSOLVE_OK) be_sync(hs[i]); });
  // problems of the cursor engine's shape go to it (one block each); the ones it hands back (status 3), and all others, run
  // on the general engine's batched launch
  std::vector<ksolve_handle*> fast, run;
  std::vector<char> alone(n, 0);   // solved on their own (results complete)
  for (uint32_t i = 0; i < n; ++i) {
    if (st[i] != KSOLVE_OK || !hs[i]->n_pods) continue;
    ksolve_handle* h = hs[i];
    if (h->fw.enabled && !h->pv.big && h->n_classes) {
      // the batched kernel is the LDS plan; a handle whose claims live in HBM (an earlier Solve() moved it there, or engine =
      // cursor-wide / cursor-hbm) runs alone on the kernel of its plan
      if (h->fw.plan.global_state == 0) fast.push_back(h);
      else { be_results_drop(&outs[i]); st[i] = solve(h, &outs[i], false); outs[i].status = st[i]; alone[i] = 1; }
    }
    else if (h->opts.engine >= 2) st[i] = fail(h, KSOLVE_ERR_UNSUPPORTED, "cursor engine requested for a problem outside its shape");
    else run.push_back(h);
  }
  if (!fast.empty()) {
    be_launch_pack_fast_batch(fast.data(), (int)fast.size());
    std::vector<int> handed_back(fast.size(), 0);
    for (size_t k = 0; k < fast.size(); ++k) {
      ksolve_handle* h = fast[k];
      int status = 0, n_claims = 0;
      be_d2h(h, &status, h->ws.status_out, 4);
      be_d2h(h, &n_claims, h->ws.n_claims_out, 4);
      be_sync(h);
      if (be_ok(h) && status != 3 && status != 1) { h->engine_used = 2; if (n_claims) be_launch_fast_records(h, n_claims); continue; }
      if (be_ok(h) && status == 3) { ks::Counters ctr{}; be_d2h(h, &ctr, h->ws.counters, sizeof(ctr)); be_sync(h); h->fast_reason = (uint32_t)ctr.cycles[20]; }
      else if (be_ok(h)) h->fast_reason = 100;
      handed_back[k] = 1;
    }
    for (size_t k = 0; k < fast.size(); ++k) {
      if (!handed_back[k]) continue;
      ksolve_handle* h = fast[k];
      uint32_t idx = 0;
      while (hs[idx] != h) ++idx;
      if (h->fast_reason == 26 && !h->opts.lds_claim_cap) {
        // more in-flight claims than the LDS plan holds: alone, on the plans that keep them in HBM (solve() escalates further)
        fast_plan_set(h, 1, h->fw.plan.rows);
        be_results_drop(&outs[idx]); st[idx] = solve(h, &outs[idx], false); outs[idx].status = st[idx]; alone[idx] = 1;
        continue;
      }
      if (h->opts.engine >= 2) { st[idx] = fail(h, KSOLVE_ERR_UNSUPPORTED, "cursor engine declined the problem (reason " + std::to_string(h->fast_reason) + ")"); continue; }
      h->fw.enabled = 0;
      st[idx] = solve_prepare(h, false);
      if (st[idx] == KSOLVE_OK) { be_sync(h); run.push_back(h); }
    }
  }
  if (!run.empty()) be_launch_pack_batch(run.data(), (int)run.size());
  parallel([&](uint32_t i) {
    if (alone[i]) return;
    if (st[i] != KSOLVE_OK) { outs[i].status = st[i]; return; }
    st[i] = solve_finish(hs[i], &outs[i]);
    if (st[i] == KSOLVE_ERR_CAPACITY && hs[i]->big_capable && !hs[i]->pv.big) { be_results_drop(&outs[i]); st[i] = solve(hs[i], &outs[i], false); }   // re-run alone on the BIG engine
    outs[i].status = st[i];   // solve_finish zeroes `out` before it can fail: the per-problem status must survive that
  });
  ksolve_status worst = KSOLVE_OK;
  for (uint32_t i = 0; i < n; ++i) if (st[i] != KSOLVE_OK && st[i] != KSOLVE_ERR_CANCELLED) worst = st[i];
  return worst;
}

static ksolve_status solve_batch_one_device(ksolve_handle** hs, uint32_t n, ksolve_results* outs);
// ksolve_solve_batch: the handles' devices side by side (one host thread each), one batch per device
static ksolve_status solve_batch(ksolve_handle** hs, uint32_t n, ksolve_results* outs) {
  std::vector<int> devices;
  for (uint32_t i = 0; i < n; ++i) { const int dv = be_device_of(hs[i]); if (std::find(devices.begin(), devices.end(), dv) == devices.end()) devices.push_back(dv); }
  if (devices.size() <= 1) return solve_batch_one_device(hs, n, outs);
  std::vector<ksolve_status> rc(devices.size(), KSOLVE_OK);
  std::vector<std::thread> pool;
  for (size_t g = 0; g < devices.size(); ++g) pool.emplace_back([&, g]() {
    std::vector<uint32_t> idx;
    std::vector<ksolve_handle*> grp;
    for (uint32_t i = 0; i < n; ++i) if (be_device_of(hs[i]) == devices[g]) { idx.push_back(i); grp.push_back(hs[i]); }
    be_thread_init(grp[0]);
    std::vector<ksolve_results> ro(grp.size());
    rc[g] = solve_batch_one_device(grp.data(), (uint32_t)grp.size(), ro.data());
    for (size_t k = 0; k < grp.size(); ++k) outs[idx[k]] = ro[k];
  });
  for (auto& th : pool) th.join();
  ksolve_status worst = KSOLVE_OK;
  for (auto r : rc) if (r != KSOLVE_OK) worst = r;
  return worst;
}

// one device: probes of a resident cluster run as one sweep per base handle, everything else as above
static ksolve_status solve_batch_one_device(ksolve_handle** hs, uint32_t n, ksolve_results* outs) {
  std::vector<uint32_t> plain_idx;
  std::vector<ksolve_handle*> bases;
  for (uint32_t i = 0; i < n; ++i) {
    if (!hs[i]->base) { plain_idx.push_back(i); continue; }
    if (std::find(bases.begin(), bases.end(), hs[i]->base) == bases.end()) bases.push_back(hs[i]->base);
  }
  if (bases.empty()) return solve_batch_plain(hs, n, outs);
  ksolve_status worst = KSOLVE_OK;
  for (ksolve_handle* b : bases) {
    std::vector<uint32_t> idx;
    std::vector<ksolve_handle*> g;
    for (uint32_t i = 0; i < n; ++i) if (hs[i]->base == b) { idx.push_back(i); g.push_back(hs[i]); }
    std::vector<ksolve_results> ro(g.size());
    std::vector<ksolve_status> st(g.size(), KSOLVE_OK);
    solve_probes(g.data(), (uint32_t)g.size(), ro.data(), st.data(), true);
    for (size_t k = 0; k < g.size(); ++k) { outs[idx[k]] = ro[k]; if (st[k] != KSOLVE_OK && st[k] != KSOLVE_ERR_CANCELLED) worst = st[k]; }
  }
  if (!plain_idx.empty()) {
    std::vector<ksolve_handle*> g;
    for (uint32_t i : plain_idx) g.push_back(hs[i]);
    std::vector<ksolve_results> ro(g.size());
    const ksolve_status rc = solve_batch_plain(g.data(), (uint32_t)g.size(), ro.data());
    for (size_t k = 0; k < g.size(); ++k) outs[plain_idx[k]] = ro[k];
    if (rc != KSOLVE_OK) worst = rc;
  }
  return worst;
}

}  // namespace ksi
