// engine.h — the pack engine: the reference's Scheduler.Solve() loop (scheduler.go:440-519) for one scheduling
// problem, executed by ONE wavefront.
//
// Why one wave: first-fit-decreasing is a serial chain — every commit changes the claim it lands on, the claim order
// (scheduler.go:598) and the NodePool limits the next pod sees — so control is wave-uniform scalar code. The 64 lanes
// are the vector unit for what IS parallel inside a step:
//   * requirement-set intersection (Requirements.Compatible / Add, requirements.go:181-197,133-140): one lane per
//     dictionary mask word, three ballots give per-key "has intersection / non-empty" for all keys at once;
//   * instance-type filtering (filterInstanceTypesByRequirements, nodeclaim.go:541-618): requirement compatibility from
//     per-(key,value) instance-type bitmasks, resource fit and offering availability one lane per instance type,
//     __ballot -> one u64 word of the surviving InstanceTypeOptions mask;
//   * first-fit selection over the in-flight claims in the reference's order with "lowest index wins"
//     (scheduler.go:667-686): ballot over the class's dead row, gather, reduce_min on (position, claim);
//   * the claim re-ordering (pdq_emul.h) and record moves.
// Latency discipline (a single wave cannot hide a dependent HBM/L2 round trip): instance-type tables, templates and
// the claim order live in LDS; a claim / class is moved as ONE contiguous record with one coalesced load (lane i moves
// word i); queue entries are fetched 64 at a time. Independent problems (NodePool components, consolidation probes)
// run as independent waves.
//
// Exact pruning: dead[class][claim] caches "CanAdd(claim, pod of this class) failed". Between two changes of a
// claim's requirement set that verdict is monotone (requests only grow, InstanceTypeOptions only shrink), so the bit
// stays valid; when a commit changes the claim's requirements the claim's column is cleared for every class.
#pragma once
#include "ksp.h"
#include "nodecheck.h"
#include <type_traits>
#include "pdq_emul.h"
#include "run_order.h"

// diagnostic counters that only the profiling build keeps (every live 64-bit counter costs the lone wave registers)
#ifdef KSOLVE_PHASE_TIMERS
#define KS_DIAG(x) x
#else
#define KS_DIAG(x) ((void)0)
#endif

namespace ks {

enum {
  E_OK = 0, E_TAINTS = 1, E_INCOMPATIBLE = 2, E_TOPOLOGY = 3, E_INSTANCE_TYPES = 4, E_RESOURCES = 5, E_NO_TEMPLATES = 6,
  E_LIMITS = 7, E_RESERVED = 8, E_EXISTING = 9, E_MIN_VALUES = 10
};

constexpr int kMaxCold = 2 * kMaxKeys + (kMaxKeys + 1) / 2;

// The engine's working set in LDS, one per wavefront. RW / IW = requirement mask words / instance-type mask words it is built for,
// CL = lines of the hot-record cache (a power of two). `Scratch` (the bounds of the library) serves every kernel but the compact
// consolidation sweep, whose wavefronts use ScratchT<32, 8, 4> when the cluster's dictionaries fit (every KWOK catalogue does):
// a third of the bytes, so that eight probes instead of four share a CU (ksolve_pack_sweep4 in ksolve.hip).
template <int RW, int IW, int CL, bool REGS = true>
struct ScratchT {
  static constexpr int kReqWords = RW, kItWords = IW, kCacheLines = CL;
  static constexpr bool kTopoLds = REGS;     // an engine on this working set may keep the topology groups' descriptors / small state in LDS (Engine::topo_to_lds); the compact sweep's never does: its engine reads them where the problem has them, with no pointers of its own to keep in registers
  static constexpr bool kRegTables = REGS;   // the engine keeps the instance-type tables in registers too (80 VGPRs; the resource-fit test of the instance-type filter without LDS traffic)
  static constexpr int kHot = RW + IW + 2 * kMaxRes + 4;
  typedef ReqBufT<RW> Buf;
  Buf merged;                       // slow-path working set (bounds / minValues / very wide dictionaries)
  Buf topo;                         // nodeRequirements ∧ topology domains (Topology.AddRequirements, topology.go:226-250)
  Buf vbase;                        // bin ∧ pod before a volume requirement alternative is added (nodeclaim.go:130-136, existingnode.go:105-106)
  uint64_t tq[RW];                  // next-domain set of one topology group (only the group key's words are used)
  int64_t gtot[kMaxRes];            // requests + the daemon overhead of the group being filtered
  uint64_t gin[IW];                 // the bin's instance types that belong to that group
  uint64_t km_old[4];               // a claim's admitted values per topology-key slot before the commit being written
  int32_t resv_cap[64];             // ReservationManager.capacity (reservationmanager.go:31)
  int dg_first[33];   // daemon-overhead groups of each template (CSR)
  uint64_t t_owned[kMaxTopoWords], t_sel[kMaxTopoWords];   // topology groups the class being placed owns / is selected by
  uint64_t t_match[kMaxTopoWords];  // getMatchingTopologies (topology.go:561-574)
  uint64_t t_active[kMaxTopoWords]; // groups created so far
  uint64_t cm[IW];                  // instance types compatible with the merged requirements
  uint64_t its[IW];                 // surviving InstanceTypeOptions
  uint64_t lim[IW];                 // instance types within NodePool limits
  uint64_t xfit[IW];                // instance types that an offering-override group makes fit (requests <= its allocatable, compatible offering)
  uint64_t xoff[IW];                // instance types with a compatible offering in an override group
  uint64_t cand[64];                // candidate list: (position << 32 | claim)
  uint64_t stage[128];              // live words of the class's dead row (two per lane: up to 8192 claims)
  int64_t total[kMaxRes];
  uint64_t claim[kHot];             // candidate bin, hot record
  uint64_t claim_cold[kMaxCold];
  uint64_t cls[RW + kMaxRes + 4];
  uint64_t cls_cold[kMaxCold];
  uint64_t out[kHot];               // record being committed
  uint64_t out_cold[kMaxCold];
  uint32_t blk_pod[64], blk_class[64], blk_last[64], blk_out[64];   // the queue block being placed: pod, class, lastLen, output index
  uint64_t tmpl_taints[32];         // template taint masks
  int64_t min_request[kMaxRes];     // min over classes per dimension (closed-claim test)
  int32_t cache_tag[CL];            // claim id held by each record-cache line, -1 = empty
  uint8_t word_key[RW];             // dictionary word -> key
};
typedef ScratchT<kMaxReqWords, kMaxItWords, 32> Scratch;
// the compact sweep's: dictionaries of <= 32 mask words, <= 512 instance types; no register tables — a probe's pods go to existing
// nodes, the instance-type filter runs for the odd new NodeClaim only, and the 80 registers are what lets two wavefronts share a SIMD
typedef ScratchT<32, 8, 4, false> ScratchSmall;

struct LdsTables {  // pointers into the dynamic LDS segment (device) / a host buffer (test emulation)
  int64_t* alloc;       // [nr][iw*64]
  uint64_t* avail;      // [iw*64]
  uint64_t* kv;         // [n_kv][iw]
  uint64_t* keymask;    // [3][nk][iw]
  uint64_t* allocok;    // [iw]
  uint16_t* kvslot;     // [rw*64]
  uint64_t* tmpl;       // [T][c_hot_words]
  uint64_t* tmpl_cold;  // [T][cold_words]
  KS_LDS uint32_t *okey, *oord, *opos;   // claim order (pdq_emul.h)
  uint64_t* closed;     // [claim words] claims that cannot take any pod any more
  uint64_t* stage_big;  // [claim words] live-set staging of the BIG engine (the others use Scratch::stage)
  KS_LDS RunTables* runs;   // BIG engine: per-count ring tables of the claim order (run_order.h), in place of okey / oord / opos
  uint64_t* cache;      // [Scratch::kCacheLines][c_hot_words] direct-mapped cache of hot claim records
  int64_t* dg_ov;       // [n_dg][nr] daemon overhead per group (scheduler.go:963-1043)
  uint64_t* dg_its;     // [n_dg][iw] instance types of the group
  char* scratch;        // the wavefront's ScratchT (the engine knows which)
  char* topo = nullptr; // room for the topology groups' descriptors and small state (LdsPlan::off_topo), or null; set by the one-problem kernels after bind
  KS_FN void bind(char* base, const LdsPlan& p, int wave = 0) {
    char* const shared = base;
    base += (size_t)wave * (size_t)p.wave_stride;   // the wavefront's own tables (compact sweep); the shared ones are re-bound below
    alloc = (int64_t*)(base + p.off_alloc); avail = (uint64_t*)(base + p.off_avail); kv = (uint64_t*)(base + p.off_kv);
    keymask = (uint64_t*)(base + p.off_keymask); allocok = (uint64_t*)(base + p.off_allocok); kvslot = (uint16_t*)(base + p.off_kvslot);
    tmpl = (uint64_t*)(base + p.off_tmpl); tmpl_cold = (uint64_t*)(base + p.off_tmplcold);
    okey = (KS_LDS uint32_t*)(base + p.off_order); oord = okey + p.order_cap; opos = oord + p.order_cap;
    runs = (KS_LDS RunTables*)(base + p.off_order);
    closed = (uint64_t*)(base + p.off_closed); cache = (uint64_t*)(base + p.off_cache);
    stage_big = (uint64_t*)(base + p.off_stage);
    dg_ov = (int64_t*)(base + p.off_dgov); dg_its = (uint64_t*)(base + p.off_dgits);
    scratch = base + p.off_scratch;
    if (wave) {
      alloc = (int64_t*)(shared + p.off_alloc); avail = (uint64_t*)(shared + p.off_avail); kv = (uint64_t*)(shared + p.off_kv);
      keymask = (uint64_t*)(shared + p.off_keymask); allocok = (uint64_t*)(shared + p.off_allocok); kvslot = (uint16_t*)(shared + p.off_kvslot);
      tmpl = (uint64_t*)(shared + p.off_tmpl); tmpl_cold = (uint64_t*)(shared + p.off_tmplcold);
      dg_ov = (int64_t*)(shared + p.off_dgov); dg_its = (uint64_t*)(shared + p.off_dgits);
    }
  }
};

// FULL = false compiles the engine for problems without topology groups, existing nodes, daemonset overhead, minValues
// and reservations (C1/C2-shaped provisioning batches): those code paths and their live state drop out of the kernel.
// BIG = true keeps the claim order in HBM and sizes the live-set staging area and the closed bitmap by the problem
// (hundreds of thousands of in-flight claims: anti-affinity / hostname-spread workloads where every pod is its own node).
template <class W, bool FULL = true, bool BIG = false, class SC = Scratch>
struct Engine {
  const ProblemView& P;
  Workspace& S;
  LdsTables L;
  SC& sc;
  const RecLayout lay;
  typedef typename SC::Buf ReqBuf;   // the requirement buffers of this engine's scratch
  static constexpr int kLineMask = SC::kCacheLines - 1;
  typedef typename std::conditional<BIG, uint32_t*, KS_LDS uint32_t*>::type order_ptr;
  typename std::conditional<BIG, RunOrder<W>, ClaimOrder<W, order_ptr>>::type order;   // BIG: one ring per pod count in HBM (run_order.h)
  int n_claims = 0;
  uint32_t host_seq = 0;
  uint32_t active_templates = 0;
  int last_err = 0, last_diag = 0;
  Counters ctr{};
  // a counter stays a wave-uniform scalar whatever it is fed with (a vector-register counter is the first thing the allocator spills, and
  // every increment of a spilled counter is a load and a store to scratch in the middle of a pod's chain)
  KS_DEV static void cadd(unsigned long long& c, unsigned long long v) { c = W::uniform(c + W::uniform(v)); }
  // topology: groups that exist so far, and the masks of the class being placed
  // Instance-type tables in registers: lane l owns instance types {j*64 + l}; with up to kRegIw mask words and kRegNr
  // resource dimensions (512 types x 4 dims: every KWOK / benchmark catalogue) the resource-fit test of
  // filterInstanceTypesByRequirements is pure VALU compares on registers, no LDS traffic. Larger problems use the LDS tables.
  static constexpr int kRegIw = 8, kRegNr = 4;
#if KS_DEVICE
  int64_t ra_[kRegIw][kRegNr];
  uint64_t rav_[kRegIw];
  KS_DEV int64_t& RA(int, int j, int r) { return ra_[j][r]; }
  KS_DEV uint64_t& RAV(int, int j) { return rav_[j]; }
#else
  int64_t ra_[64][kRegIw][kRegNr];
  uint64_t rav_[64][kRegIw];
  int64_t& RA(int l, int j, int r) { return ra_[l][j][r]; }
  uint64_t& RAV(int l, int j) { return rav_[l][j]; }
#endif
  bool regs_ok = false;
  // The topology groups' descriptors and their small mutable state (registered domains, per-domain counts, non-empty-domain
  // counts) are read through these pointers — the problem's HBM tables, or copies in LDS when the launch planned room for them
  // (LdsPlan::off_topo) and the kernel allows it (one problem per kernel; probes and batches keep them in HBM). Measured in round 4
  // as an experiment (1.08-1.09x on the configs[2] shape: a tenth of the 136 dependent vector reads per pod), shipped in round 5.
  struct TopoPtrs { const uint8_t* type; const int32_t* key; const int8_t* key_slot; const int16_t* host_slot; const int32_t *max_skew, *min_domains;
                    const uint8_t *f_affinity, *f_taint; const uint32_t* f_first; const uint64_t* f_tolerates; } TT;
  uint64_t* tgD = nullptr; int32_t* tgC = nullptr; int32_t* tgN = nullptr;
  KS_DEV const uint8_t* tt_type() const { if constexpr (SC::kTopoLds) return TT.type; else return P.topo.type; }
  KS_DEV const int32_t* tt_key() const { if constexpr (SC::kTopoLds) return TT.key; else return P.topo.key; }
  KS_DEV const int8_t* tt_key_slot() const { if constexpr (SC::kTopoLds) return TT.key_slot; else return P.topo.key_slot; }
  KS_DEV const int16_t* tt_host_slot() const { if constexpr (SC::kTopoLds) return TT.host_slot; else return P.topo.host_slot; }
  KS_DEV const int32_t* tt_max_skew() const { if constexpr (SC::kTopoLds) return TT.max_skew; else return P.topo.max_skew; }
  KS_DEV const int32_t* tt_min_domains() const { if constexpr (SC::kTopoLds) return TT.min_domains; else return P.topo.min_domains; }
  KS_DEV const uint8_t* tt_f_affinity() const { if constexpr (SC::kTopoLds) return TT.f_affinity; else return P.topo.f_affinity; }
  KS_DEV const uint8_t* tt_f_taint() const { if constexpr (SC::kTopoLds) return TT.f_taint; else return P.topo.f_taint; }
  KS_DEV const uint32_t* tt_f_first() const { if constexpr (SC::kTopoLds) return TT.f_first; else return P.topo.f_first; }
  KS_DEV const uint64_t* tt_f_tolerates() const { if constexpr (SC::kTopoLds) return TT.f_tolerates; else return P.topo.f_tolerates; }
  KS_DEV uint64_t* tg_D() const { if constexpr (SC::kTopoLds) return tgD; else return S.tg_domains; }
  KS_DEV int32_t* tg_C() const { if constexpr (SC::kTopoLds) return tgC; else return S.tg_counts; }
  KS_DEV int32_t* tg_N() const { if constexpr (SC::kTopoLds) return tgN; else return S.tg_nonzero; }
  uint64_t pending_reserved = 0;    // offerings the last successful can_add wants reserved (offeringsToReserve, nodeclaim.go:303-350)
  bool min_values_best_effort = false;   // MinValuesPolicyBestEffort (scheduler.go:117)
  bool minv_lowered = false;        // the last can_add lowered a minValues requirement (BestEffort)
  bool cur_M = false;               // the class being placed has matching topology groups
  bool cur_rec = false;             // ... or is counted by some group when it is committed
  int cur_class = 0;
  uint64_t cur_hp_use = 0, cur_hp_conf = 0;   // host-port triples the pod being placed binds / that match one of them
  uint32_t cur_vol_first = 0, cur_vol_n = 0;  // volume requirement alternatives of the pod being placed (PodData.VolumeRequirements)
  uint64_t bin_hp = 0;                        // host-port triples already bound by the pods of the candidate bin
  bool topo_reached = false;        // the last can_add got as far as the topology stage (its verdict is not cacheable)
  uint32_t cur_pv_first = 0, cur_pv_n = 0;    // the pod's volumes of drivers with a limit somewhere (ProblemView::pod_pvs)
  int n_pv_log = 0;                 // entries of Workspace::pv_log
  int cur_out = 0;                  // where the pod being placed reports its result: its pod index, or its position in Workspace::pr_sorted (probes)
  int n_revived = 0;                // probes: entries of Workspace::pr_revived
  // (round 6) a probe's overlay index in a vector register: with ov_cap == 64 (a probe of up to 32 pods — every single-node probe) lane s
  // holds the node of overlay slot s (node + 1, slots handed out in order), so "is node e overlaid, and where" is a walk over n_ov
  // readlanes instead of a hash probe into HBM — it was asked three times per pod (the block's lanes, node_merge, ov_touch), each a
  // dependent L2 round trip of the probe's chain. Larger probes keep the open-addressing table in Workspace::ov_key.
  LaneVar<uint32_t> ovk_;
  int n_ov = 0;
  bool ov_regs = false;
  // a probe's removed nodes (ascending; 0xFFFFFFFF behind the last) in two vector registers when there are at most 128 of them — a multi-node
  // prefix removes up to 101: every scan step and every evaluation count walked the list in HBM, a dependent load per entry
  LaneVar<uint32_t> prm0_, prm1_;
  bool prm_regs = false;
  bool cur_exempt = false;          // the pod being placed is pending or comes from a deleting node (scheduler.go:628 does not skip nodes for it)

  KS_DEV Engine(const ProblemView& p, Workspace& s, const LdsTables& l) : P(p), S(s), L(l), sc(*(SC*)l.scratch), lay(p.lay) {
    if constexpr (BIG) order.init(L.runs, s.o_ring, s.o_cnt, s.o_pos, s.o_key, s.o_ord, s.run_tabs, s.run_off, s.run_log, s.run_kmax);
    else { order.key = L.okey; order.ord = L.oord; order.pos = L.opos; }
    min_values_best_effort = s.min_values_best_effort != 0;
    const TopoView& T0 = p.topo;
    TT.type = T0.type; TT.key = T0.key; TT.key_slot = T0.key_slot; TT.host_slot = T0.host_slot; TT.max_skew = T0.max_skew; TT.min_domains = T0.min_domains;
    TT.f_affinity = T0.f_affinity; TT.f_taint = T0.f_taint; TT.f_first = T0.f_first; TT.f_tolerates = T0.f_tolerates;
    tgD = s.tg_domains; tgC = s.tg_counts; tgN = s.tg_nonzero;
  }
  // copies the descriptors into LDS and points the mutable group state there (before solve() fills it from the pristine tables)
  KS_DEV void topo_to_lds() {
    if (!SC::kTopoLds || !FULL || !L.topo || !P.topo.n_groups || S.probe) return;
    const TopoView& T0 = P.topo;
    const int G = T0.n_groups, dw = T0.dom_words;
    char* q = L.topo;
    auto al = [](size_t b) { return (b + 7) & ~(size_t)7; };
    { uint8_t* d = (uint8_t*)q; const uint8_t* a = T0.type; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.type = d; q += al(G); }
    { int32_t* d = (int32_t*)q; const int32_t* a = T0.key; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.key = d; q += al((size_t)G * 4); }
    { int8_t* d = (int8_t*)q; const int8_t* a = T0.key_slot; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.key_slot = d; q += al(G); }
    { int16_t* d = (int16_t*)q; const int16_t* a = T0.host_slot; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.host_slot = d; q += al((size_t)G * 2); }
    { int32_t* d = (int32_t*)q; const int32_t* a = T0.max_skew; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.max_skew = d; q += al((size_t)G * 4); }
    { int32_t* d = (int32_t*)q; const int32_t* a = T0.min_domains; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.min_domains = d; q += al((size_t)G * 4); }
    { uint8_t* d = (uint8_t*)q; const uint8_t* a = T0.f_affinity; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.f_affinity = d; q += al(G); }
    { uint8_t* d = (uint8_t*)q; const uint8_t* a = T0.f_taint; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.f_taint = d; q += al(G); }
    { uint32_t* d = (uint32_t*)q; const uint32_t* a = T0.f_first; W::for_n(G + 1, [&](int i) { d[i] = a[i]; }); TT.f_first = d; q += al((size_t)(G + 1) * 4); }
    { uint64_t* d = (uint64_t*)q; const uint64_t* a = T0.f_tolerates; W::for_n(G, [&](int i) { d[i] = a[i]; }); TT.f_tolerates = d; q += al((size_t)G * 8); }
    tgD = (uint64_t*)q; q += al((size_t)G * dw * 8);
    tgC = (int32_t*)q; q += al((size_t)G * dw * 64 * 4);
    tgN = (int32_t*)q; q += al((size_t)G * 4);
    W::sync();
  }

  // ------------------------------------------------------------------------------------------------------------
  // record helpers
  KS_DEV void load_words(uint64_t* dst, const uint64_t* src, int n) {
    W::for_n(n, [&](int i) { dst[i] = src[i]; });
  }
  KS_DEV static uint32_t lo32(uint64_t v) { return (uint32_t)v; }
  KS_DEV static uint32_t hi32(uint64_t v) { return (uint32_t)(v >> 32); }
  KS_DEV ReqRef claim_ref(const uint64_t* hot, const uint64_t* cold) const {
    ReqRef r;
    uint64_t f0 = hot[lay.c_f0()], f1 = hot[lay.c_f1()];
    r.mask = hot + lay.c_mask(); r.defined = lo32(f0); r.complement = hi32(f0); r.has_gte = lo32(f1); r.has_lte = hi32(f1);
    r.gte = (const int64_t*)cold; r.lte = (const int64_t*)(cold + lay.nk);
    r.minv = (hi32(hot[lay.c_meta2()]) & 2u) ? (const int32_t*)(cold + 2 * lay.nk) : nullptr;
    return r;
  }
  KS_DEV ReqRef class_ref(const uint64_t* hot, const uint64_t* cold) const {
    ReqRef r;
    uint64_t f0 = hot[lay.k_f0()], f1 = hot[lay.k_f1()];
    r.mask = hot + lay.k_mask(); r.defined = lo32(f0); r.complement = hi32(f0); r.has_gte = lo32(f1); r.has_lte = hi32(f1);
    r.gte = (const int64_t*)cold; r.lte = (const int64_t*)(cold + lay.nk);
    r.minv = (lo32(hot[lay.k_meta()]) & 1u) ? (const int32_t*)(cold + 2 * lay.nk) : nullptr;
    return r;
  }

  // ------------------------------------------------------------------------------------------------------------
  // one-time: instance-type tables, dictionary helpers and templates into LDS. Two halves: the tables every solve of the same
  // problem reads alike (`shared`: the compact sweep fills them once per workgroup and its wavefronts share them) and the
  // wavefront's own working set (Scratch, cache tags, the closed bitmap, the register tables).
  KS_DEV void load_tables(bool shared = true) {
    const Dict& d = P.dict;
    const int nr = P.n_res, iw = P.it_words, n_its = P.n_its, np = iw * 64;
    const ProblemView& Pv = P;
    LdsTables& Lt = L;
    if (shared) {
      for (int r = 0; r < nr; ++r)
        W::for_n(np, [&](int it) { Lt.alloc[(size_t)r * np + it] = it < n_its ? Pv.it_alloc[(size_t)r * n_its + it] : INT64_MIN; });
      W::for_n(np, [&](int it) { Lt.avail[it] = it < n_its ? Pv.it_base_avail[it] : 0; });   // base group == every offering unless overrides exist
      W::for_n(iw, [&](int w) { Lt.allocok[w] = Pv.it_alloc_ok[w]; });
      const int nki = d.n_keys * iw;
      W::for_n(3 * nki, [&](int i) {
        int which = i / nki, rest = i % nki;
        const uint64_t* src = which == 0 ? Pv.key_undef : which == 1 ? Pv.key_compl : Pv.key_neg;
        Lt.keymask[i] = src[rest];
      });
      W::for_n(d.req_words * 64, [&](int v) {
        uint16_t slot = Pv.kv_slot[v];
        Lt.kvslot[v] = slot;
        if (slot != 0xFFFF) for (int w = 0; w < iw; ++w) Lt.kv[(size_t)slot * iw + w] = Pv.kv_has[(size_t)v * iw + w];
      });
      W::for_n(Pv.n_dg * nr, [&](int i) { Lt.dg_ov[i] = Pv.dg_ov[i]; });
      W::for_n(Pv.n_dg * iw, [&](int i) { Lt.dg_its[i] = Pv.dg_its[i]; });
    }
    uint8_t* wk = sc.word_key;
    if (W::leader())
      for (int k = 0; k < d.n_keys; ++k)
        for (uint32_t w = d.key_word_off[k]; w < d.key_word_off[k + 1]; ++w) wk[w] = (uint8_t)k;
    uint64_t* tt = sc.tmpl_taints; int64_t* mr = sc.min_request; int32_t* tag = sc.cache_tag;
    W::for_n(32, [&](int t) { tt[t] = t < Pv.n_templates ? Pv.tmpl_taints[t] : 0; if (t < SC::kCacheLines) tag[t] = -1; });
    W::for_n(nr, [&](int r) { mr[r] = Pv.min_request[r]; });
    W::for_n(BIG ? Pv.lds.stage_words : (S.probe ? S.pr_order_cap : Pv.lds.order_cap) / 64, [&](int w) { Lt.closed[w] = 0; });
    W::sync();
    W::for_n(Pv.n_templates + 1, [&](int t) { sc.dg_first[t] = Pv.dg_first[t]; });
    regs_ok = SC::kRegTables && (!FULL || (iw <= kRegIw && nr <= kRegNr && Pv.n_xg == 0));   // lite problems fit the register tables by definition
    if (SC::kRegTables && regs_ok) {
      W::ballot([&](int l) {
#pragma unroll
        for (int j = 0; j < kRegIw; ++j) {
          const int it = j * 64 + l;
          RAV(l, j) = (j < iw && it < n_its) ? Pv.it_base_avail[it] : 0ull;
#pragma unroll
          for (int r = 0; r < kRegNr; ++r) RA(l, j, r) = (j < iw && r < nr && it < n_its) ? Pv.it_alloc[(size_t)r * n_its + it] : INT64_MIN;
        }
        return false;
      });
    }
  }

  // ------------------------------------------------------------------------------------------------------------
  // Instance types compatible with a requirement set: InstanceType.Requirements.Intersects(reqs) == nil for every
  // type at once (nodeclaim.go:620-622, requirements.go:254-274). Only keys that some instance type defines can
  // exclude a type.
  KS_DEV void compat_mask(const ReqRef& q) {
    const Dict& d = P.dict;
    const LdsTables& Lt = L;
    uint64_t* cm = sc.cm;
    const uint64_t* m = q.mask;
    const int iw = P.it_words, nk = d.n_keys;
    const uint32_t keys0 = q.defined & (P.it_keys | (d.key_it >= 0 ? (1u << d.key_it) : 0u));
    W::for_n(iw, [&](int w) {
      uint64_t acc = ~0ull;
      uint32_t keys = keys0;
      while (keys) {
        int k = __builtin_ctz(keys);
        keys &= keys - 1;
        uint32_t w0 = d.key_word_off[k], w1 = d.key_word_off[k + 1];
        bool comp = bit(q.complement, k);
        bool hg = bit(q.has_gte, k), hl = bit(q.has_lte, k);
        int64_t g = hg ? q.gte[k] : 0, l = hl ? q.lte[k] : 0;
        if (k == d.key_it) {
          // the instance-type key's dictionary IS the instance-type list and every type requires In [own name]
          uint64_t mm = m[w0 + w];
          acc &= comp ? inbounds_word(d, w0 + w, ~mm, hg, g, hl, l) : mm;
          continue;
        }
        uint64_t r = Lt.keymask[(size_t)k * iw + w];                                  // key_undef
        bool nonempty = false;
        if (!comp) {
          for (uint32_t x = w0; x < w1; ++x) {
            uint64_t bits = m[x];
            if (bits) nonempty = true;
            while (bits) { int b = ctz64(bits); bits &= bits - 1; uint16_t s = Lt.kvslot[x * 64 + b]; if (s != 0xFFFF) r |= Lt.kv[(size_t)s * iw + w]; }
          }
          if (!nonempty) r |= Lt.keymask[(size_t)(2 * nk + k) * iw + w];             // DoesNotExist vs {NotIn, DoesNotExist}: requirements.go:260-265
        } else {
          r |= Lt.keymask[(size_t)(nk + k) * iw + w];                                 // two complements always intersect: requirement.go:226-228
          for (uint32_t x = w0; x < w1; ++x) {
            if (m[x]) nonempty = true;
            uint64_t bits = inbounds_word(d, x, ~m[x] & d.value_valid[x], hg, g, hl, l);
            while (bits) { int b = ctz64(bits); bits &= bits - 1; uint16_t s = Lt.kvslot[x * 64 + b]; if (s != 0xFFFF) r |= Lt.kv[(size_t)s * iw + w]; }
          }
          if (nonempty) r |= Lt.keymask[(size_t)(2 * nk + k) * iw + w];              // NotIn vs {NotIn, DoesNotExist}
        }
        acc &= r;
      }
      cm[w] = acc;
    });
  }

  // zone x capacity-type cells an offering may sit in to be compatible with the requirements (types.go:553-570:
  // reqs.IsCompatible(offering.Requirements, AllowUndefinedWellKnownLabels); offerings carry single In values).
  KS_DEV uint64_t offering_cells(const ReqRef& r) {
    const Dict& d = P.dict;
    uint32_t zones = 0, cts = 0;
    if (d.key_zone >= 0 && bit(r.defined, d.key_zone)) {
      for (int z = 0; z < P.n_zones; ++z) if (req_has(d, r, d.key_zone, d.key_word_off[d.key_zone], z)) zones |= 1u << z;
    } else zones = (1u << P.n_zones) - 1;
    if (d.key_ct >= 0 && bit(r.defined, d.key_ct)) {
      for (int c = 0; c < P.n_cts; ++c) if (req_has(d, r, d.key_ct, d.key_word_off[d.key_ct], c)) cts |= 1u << c;
    } else cts = (1u << P.n_cts) - 1;
    uint64_t cells = 0;
    while (zones) { int z = __builtin_ctz(zones); zones &= zones - 1; cells |= (uint64_t)cts << (z * 4); }
    return cells;
  }

  // filterInstanceTypesByRequirements for one candidate bin: its' = its ∩ compatible ∩ fits ∩ hasOffering.
  // `full` = the requirement set differs from the bin's own, so compatibility and offerings must be re-evaluated; when
  // the requirements are unchanged the bin's instance types already satisfy both (they were filtered by these very
  // requirements at the last commit, nodeclaim.go:250-253) and only the resource fit can drop types.
  // `tmpl` selects the daemon-overhead groups (instance types of a template that share the same set of compatible
  // daemonset pods, scheduler.go:963-1043): an instance type must fit requests + its group's overhead
  // (nodeclaim.go:558-566). tmpl < 0: no overhead (NewScheduler's prefilter, scheduler.go:159).
  struct FilterDiagAcc { bool d_req = false, d_fit = false, d_off = false, d_ro = false, d_fo = false; };
  KS_DEV bool filter_instance_types(const uint64_t* bin_its, const int64_t* total, bool full, const ReqRef& reqs, bool want_diag, int tmpl) {
    uint64_t cells = ~0ull;
    if (full) {
      KS_DIAG(cadd(ctr.full_filters, 1));
      compat_mask(reqs);
      cells = offering_cells(reqs);
    } else if (FULL && P.n_xg) {
      cells = offering_cells(reqs);   // which GROUP of a type has the compatible offering decides which allocatable counts
    }
    FilterDiagAcc acc;
    const int nr = P.n_res, iw = P.it_words;
    const int g0 = (!FULL || tmpl < 0) ? 0 : sc.dg_first[tmpl], g1 = (!FULL || tmpl < 0) ? 1 : sc.dg_first[tmpl + 1];
    // a daemon-overhead group whose host ports — its daemon pods' and the bin's own pods' (nodeclaim.go:256-259) — match
    // one of the pod's is skipped as a whole (nodeclaim.go:562-565); the NewScheduler prefilter (tmpl < 0) has no pod
    const bool hp = FULL && tmpl >= 0 && P.hp_on && cur_hp_conf != 0;
    bool any;
    if (hp && g1 - g0 == 1 && ((bin_hp | P.dg_hp[g0]) & cur_hp_conf)) {
      uint64_t* its_out = sc.its;
      W::for_n(iw, [&](int w) { its_out[w] = 0; });
      W::sync();
      any = false;
    } else if (!FULL || g1 - g0 == 1) {
      const int64_t* tot = total;
      if (FULL && tmpl >= 0 && ((P.dg_nonzero >> g0) & 1)) {
        const int64_t* ov = L.dg_ov + (size_t)g0 * nr;
        int64_t* gt = sc.gtot;
        W::for_n(nr, [&](int r) { gt[r] = total[r] + ov[r]; });
        tot = sc.gtot;
      }
      any = filter_core(bin_its, tot, full, cells, want_diag, false, acc);
    } else {
      uint64_t* its_out = sc.its;
      W::for_n(iw, [&](int w) { its_out[w] = 0; });
      any = false;
      for (int g = g0; g < g1; ++g) {
        if (hp && ((bin_hp | P.dg_hp[g]) & cur_hp_conf)) continue;
        const uint64_t* gm = L.dg_its + (size_t)g * iw;
        const int64_t* ov = L.dg_ov + (size_t)g * nr;
        uint64_t* gin = sc.gin;
        const uint64_t some = W::ballot([&](int w) { if (w >= iw) return false; const uint64_t v = bin_its[w] & gm[w]; gin[w] = v; return v != 0; });
        if (!some) continue;
        int64_t* gt = sc.gtot;
        W::for_n(nr, [&](int r) { gt[r] = total[r] + ov[r]; });
        any = filter_core(sc.gin, sc.gtot, full, cells, want_diag, true, acc) || any;
      }
    }
    if (want_diag) last_diag = (acc.d_req ? 1 : 0) | (acc.d_fit ? 2 : 0) | (acc.d_off ? 4 : 0) | (acc.d_ro ? 16 : 0) | (acc.d_fo ? 32 : 0);
    return any;
  }
  // one group: its_out (sc.its) = / |= in ∩ compatible ∩ fits ∩ hasOffering
  KS_DEV bool filter_core(const uint64_t* bin_its, const int64_t* total, bool full, uint64_t cells, bool want_diag, bool accumulate, FilterDiagAcc& acc) {
    const LdsTables& Lt = L;
    const int nr = P.n_res, iw = P.it_words, np = iw * 64;
    uint64_t any = 0;
    if (SC::kRegTables && regs_ok && !want_diag) {
      // register tables: every compare of the step is VALU work on this lane's own instance types
      unsigned long long q0 = W::clock();
      int64_t tt[kRegNr];
#pragma unroll
      for (int r = 0; r < kRegNr; ++r) tt[r] = r < nr ? total[r] : INT64_MIN;
      uint64_t* its_out = sc.its;
      const uint64_t* cmw = sc.cm;
      // eight fit ballots (words past it_words hold padding and are ignored below), each written into its lane of one
      // vector register; then ONE round of LDS traffic: lane j combines word j (its ∩ allocatable-ok ∩ compatible ∩ fits)
      LaneVec64 fitv;
      W::ballots8(8, [&](int l, int j) {
        int f = full ? (int)((RAV(l, j) & cells) != 0) : 1;
#pragma unroll
        for (int r = 0; r < kRegNr; ++r) f &= (int)(tt[r] <= RA(l, j, r));
        return f != 0;
      }, [&](int j, uint64_t fit_and_off) { fitv.set(j, fit_and_off); });
      unsigned long long q1 = W::clock();
      ctr.cycles[15] += q1 - q0;
      const uint64_t nonempty = W::ballot([&](int l) {
        if (l >= iw) return false;
        const uint64_t fw = fitv.get(l);
        const uint64_t keep = (full ? cmw[l] : ~0ull) & bin_its[l] & Lt.allocok[l] & fw;
        its_out[l] = accumulate ? (its_out[l] | keep) : keep;
        return keep != 0;
      });
      W::sync();
      ctr.cycles[16] += W::clock() - q1;
      return nonempty != 0;
    }
    bool& d_req = acc.d_req; bool& d_fit = acc.d_fit; bool& d_off = acc.d_off; bool& d_ro = acc.d_ro; bool& d_fo = acc.d_fo;
    // offering-override groups (types.go:202-269): one lane per mask word collects the types that an EXTRA group lets
    // through — requests within the group's allocatable (no negative dimension, resources.go:190) and an offering of
    // that group compatible with the requirements (nodeclaim.go:624-638) — and the types with a compatible offering in one
    const int nx = FULL ? P.n_xg : 0;
    const uint64_t* xf = sc.xfit;
    const uint64_t* xo = sc.xoff;
    if (nx) {
      uint64_t* wf = sc.xfit; uint64_t* wo = sc.xoff;
      const ProblemView& Pv = P;
      W::for_n(iw, [&](int w) {
        uint64_t f = 0, o = 0;
        for (int e = 0; e < nx; ++e) {
          const uint32_t it = Pv.xg_it[e];
          if ((int)(it >> 6) != w || !(Pv.xg_avail[e] & cells)) continue;
          bool fit = true;
          for (int r = 0; r < nr; ++r) { const int64_t a = Pv.xg_alloc[(size_t)r * nx + e]; fit = fit && a >= 0 && total[r] <= a; }
          o |= 1ull << (it & 63);
          if (fit) f |= 1ull << (it & 63);
        }
        wf[w] = f; wo[w] = o;
      });
      W::sync();
    }
    const bool offer = full || nx != 0;
    for (int w0 = 0; w0 < iw; w0 += 8) {
      const int n = iw - w0 < 8 ? iw - w0 : 8;
      // one lane per instance type, eight mask words per step; every allocatable / availability load of the step is in
      // flight at once (resource fit AND a compatible offering: fits() reports itFits only with one, nodeclaim.go:624-638)
      uint64_t* its_out = sc.its;
      const uint64_t* cmw = sc.cm;
      W::ballots8(n, [&](int l, int j) {
        int it = (w0 + j) * 64 + l;
        int f = offer ? (int)((Lt.avail[it] & cells) != 0) : 1;
        for (int r = 0; r < nr; ++r) f &= (int)(total[r] <= Lt.alloc[(size_t)r * np + it]);
        return f != 0;
      }, [&](int j, uint64_t fit_and_off) {
        const int w = w0 + j;
        const uint64_t in = bin_its[w];
        const uint64_t cm = full ? cmw[w] : ~0ull;
        const uint64_t itfits = in & ((Lt.allocok[w] & fit_and_off) | (nx ? xf[w] : 0ull));
        const uint64_t keep = cm & itfits;
        KS_DIAG(cadd(ctr.it_evaluations, (unsigned long long)(popc64(in))));
        if (want_diag) { d_req |= (in & cm) != 0; d_fit |= itfits != 0; d_fo |= (itfits & ~cm) != 0; }
        W::store(&its_out[w], (uint64_t)(accumulate ? (its_out[w] | keep) : keep));
        any |= keep;
      });
      if (want_diag) {
        // InstanceTypeFilterError flags (nodeclaim.go:585-592) need resource fit and offering separately (failure path only)
        W::ballots8(n, [&](int l, int j) { return (Lt.avail[(w0 + j) * 64 + l] & cells) != 0; },
                    [&](int j, uint64_t offb) { W::store(&sc.lim[w0 + j], (uint64_t)(offb | (nx ? xo[w0 + j] : 0ull))); });
        W::sync();
        W::ballots8(n, [&](int l, int j) {
          int it = (w0 + j) * 64 + l;
          int f = nx ? (int)((Lt.avail[it] & cells) != 0) : 1;    // with override groups the base allocatable only counts with a base offering
          for (int r = 0; r < nr; ++r) f &= (int)(total[r] <= Lt.alloc[(size_t)r * np + it]);
          return f != 0;
        }, [&](int j, uint64_t fitb) {
          const int w = w0 + j;
          const uint64_t in = bin_its[w];
          const uint64_t cm = full ? cmw[w] : ~0ull;
          const uint64_t off = offer ? (in & sc.lim[w]) : in;
          const uint64_t itfits = (in & Lt.allocok[w] & fitb & off) | (nx ? (in & xf[w]) : 0ull);
          d_off |= off != 0;
          d_ro |= (in & cm & off & ~itfits) != 0;
        });
      }
    }
    W::sync();
    return any != 0;
  }

  // ------------------------------------------------------------------------------------------------------------
  // Requirements.Compatible + Add for a bin record against the class record in sc.cls, both without bounds, one lane
  // per dictionary word. Writes the merged masks + flag words into sc.out. Returns false when incompatible.
  KS_DEV bool fast_compat_merge(const uint64_t* bin, uint32_t kdef, bool* changed) {
    const Dict& d = P.dict;
    const int rw = lay.rw;
    const uint64_t f0 = bin[lay.c_f0()];
    const uint32_t cdef = lo32(f0), ccomp = hi32(f0);
    const uint32_t kcomp = hi32(sc.cls[lay.k_f0()]) & kdef;
    const uint64_t* a = bin + lay.c_mask();
    const uint64_t* b = sc.cls + lay.k_mask();
    const uint8_t* wk = sc.word_key;
    const uint32_t both = cdef & kdef;
    uint32_t undef = kdef & ~cdef & ~d.well_known_mask;
    // One pass over the words: non-empty (bin), non-empty (pod), has-intersection, and — optimistically, it is only used
    // when the sets turn out compatible — the merged word (Requirement.Intersection, requirement.go:181-214 without
    // bounds) and whether it differs from the bin's.
    uint64_t NA = 0, NB = 0, H = 0, diff = 0;
    uint64_t* o = sc.out + lay.c_mask();
    W::ballot4([&](int l) {
      if (l >= rw) return 0;
      const int k = wk[l];
      const bool ca = (ccomp >> k) & 1, cb = (kcomp >> k) & 1;
      const uint64_t x = a[l], y = b[l];
      const uint64_t inter = ca ? (cb ? ~0ull : (y & ~x)) : (cb ? (x & ~y) : (x & y));
      uint64_t v = x;
      if ((kdef >> k) & 1) v = !((cdef >> k) & 1) ? y : (ca ? (cb ? (x | y) : (y & ~x)) : (cb ? (x & ~y) : (x & y)));
      o[l] = v;
      return (x != 0 ? 1 : 0) | (y != 0 ? 2 : 0) | (inter != 0 ? 4 : 0) | (v != x ? 8 : 0);
    }, NA, NB, H, diff);
    auto lanes_of = [&](int k) { uint32_t w0 = d.key_word_off[k], n = d.key_word_off[k + 1] - w0; return (n >= 64 ? ~0ull : ((1ull << n) - 1)) << w0; };
    while (undef) {  // a key the bin does not define: only NotIn / DoesNotExist may pass (requirements.go:185-193)
      int k = __builtin_ctz(undef);
      undef &= undef - 1;
      bool ne = (NB & lanes_of(k)) != 0;
      bool comp = (kcomp >> k) & 1;
      if (!(comp ? ne : !ne)) return false;
    }
    uint32_t bk = both;
    while (bk) {  // Intersects — requirements.go:254-274
      int k = __builtin_ctz(bk);
      bk &= bk - 1;
      uint64_t ln = lanes_of(k);
      bool ca = (ccomp >> k) & 1, cb = (kcomp >> k) & 1;
      if ((ca && cb) || (H & ln)) continue;
      bool nea = (NA & ln) != 0, neb = (NB & ln) != 0;
      bool nega = ca ? nea : !nea, negb = cb ? neb : !neb;
      if (nega && negb) continue;
      return false;
    }
    const uint32_t ndef = cdef | kdef;
    const uint32_t ncomp = (ccomp & ~kdef) | (ccomp & kcomp & both) | (kcomp & ~cdef);
    if (W::leader()) { sc.out[lay.c_f0()] = (uint64_t)ndef | ((uint64_t)ncomp << 32); sc.out[lay.c_f1()] = 0; }
    W::sync();
    *changed = diff != 0 || ndef != cdef || ncomp != ccomp;
    return true;
  }

  // ------------------------------------------------------------------------------------------------------------
  // Topology (topology.go, topologygroup.go, topologynodefilter.go). A group keyed on a dictionary key keeps one
  // counter per dictionary value; a kubernetes.io/hostname group keeps one counter per bin (existing node / claim):
  // a bin's hostname requirement is always the single value In [its own hostname] (nodeclaim.go:97, existingnode.go:72),
  // so only the single-domain branches of nextDomain* (topologygroup.go:236-249,331-343,407-415) can run for it.
  // One lane per domain value: Has() tests, min/arg-min over counters and the resulting In-set are ballots/reductions.
  static constexpr uint64_t NONE = ~(uint64_t)0;
  KS_DEV bool topo_has(const ReqRef& r, int key, uint32_t w, int b) const {   // r.Get(key).Has(value); undefined key = Exists
    if (!bit(r.defined, key)) return true;
    return req_has(P.dict, r, key, w, b);
  }
  // podDomains.Has(bin's hostname)
  KS_DEV bool pod_has_host(const ReqRef& pod, int bin_kind, int bin) const {
    const int hn = P.dict.key_hostname;
    if (hn < 0 || !bit(pod.defined, hn)) return true;
    const bool open = bit(pod.complement, hn) && !bit(pod.has_gte, hn) && !bit(pod.has_lte, hn);
    if (bin_kind == 0) return open;   // hostname-placeholder-N: outside every dictionary and not an integer
    const int v = P.topo.node_host_value[bin];
    if (v < 0) return open;
    return req_has(P.dict, pod, hn, P.dict.key_word_off[hn] + (uint32_t)(v >> 6), v & 63);
  }
  // Pods a hostname group counts on existing node e. Outside probes: the solve's own table. In a probe the cluster's counts stay
  // SHARED and pristine (TopoView::node_counts0) like the node tables; the probe keeps what its commits add in the overlay slot of
  // the node (tg_node_counts is [n_host_groups][ov_cap] there, zero at the start) — a single-node probe touches a handful of 100k
  // nodes, and copying a counter per node and group into every probe made a 10k-probe sweep of a cluster with hostname groups
  // 12 GB of arena. A removed node counts nothing: its pods are the displaced ones. Per lane.
  KS_DEV int32_t node_host_count(int hs, int e) const {
    if (!S.probe) return S.tg_node_counts[(size_t)hs * P.n_nodes + e];
    if (probe_node_removed(e)) return 0;
    int32_t c = P.topo.node_counts0[(size_t)hs * P.n_nodes + e];
    const int os = ov_find(e);
    if (os >= 0) c += S.tg_node_counts[(size_t)hs * S.ov_cap + os];
    return c;
  }
  KS_DEV int32_t host_count(int g, int bin_kind, int bin) const {
    const int hs = tt_host_slot()[g];
    return bin_kind == 0 ? S.tg_claim_counts[(size_t)hs * S.max_claims + bin] : node_host_count(hs, bin);
  }
  // Record: one more pod on the bin; returns the count before it (uniform)
  KS_DEV int32_t host_count_add(int g, int bin_kind, int bin) {
    const int hs = tt_host_slot()[g];
    if (bin_kind == 0) { int32_t* p = S.tg_claim_counts + (size_t)hs * S.max_claims + bin; const int32_t c = *p; W::store(p, c + 1); return c; }
    if (!S.probe) { int32_t* p = S.tg_node_counts + (size_t)hs * P.n_nodes + bin; const int32_t c = *p; W::store(p, c + 1); return c; }
    const int32_t before = node_host_count(hs, bin);
    const int os = ov_touch(bin);    // the commit that records has touched the node already
    int32_t* p = S.tg_node_counts + (size_t)hs * S.ov_cap + os;
    W::store(p, *p + 1);
    return before;
  }
  // anyCompatiblePodDomain for a hostname group — topologygroup.go:393-400
  KS_DEV bool topo_any_compatible_host(int g, const ReqRef& pod) {
    const int hs = tt_host_slot()[g];
    const int32_t* cc = S.tg_claim_counts + (size_t)hs * S.max_claims;
    const int ne = P.n_nodes, ncl = n_claims;
    if (W::find_first(0, ne, [&](int e) { return node_host_count(hs, e) > 0 && pod_has_host(pod, 1, e); }) < ne) return true;
    if (!pod_has_host(pod, 0, 0)) return false;
    return W::find_first(0, ncl, [&](int c) { return cc[c] > 0; }) < ncl;
  }
  // the hostname groups of the class on bin `bin` (-1 = a claim that does not exist yet: count 0)
  KS_DEV bool topo_hostname_ok(int g, bool self, const ReqRef& pod, int bin_kind, int bin) {
    const TopoView& T = P.topo;
    const int cnt = bin >= 0 ? host_count(g, bin_kind, bin) : 0;
    switch (tt_type()[g]) {
      case 0: return (long long)cnt + (self ? 1 : 0) <= (long long)tt_max_skew()[g];       // topologygroup.go:240-247
      case 2: return cnt == 0;                                                            // :409-414
      default:                                                                            // :331-343
        if (!pod_has_host(pod, bin_kind, bin)) return false;
        if (cnt > 0) return true;
        if (!self) return false;
        if (tg_N()[g] == 0) return true;
        return !topo_any_compatible_host(g, pod);
    }
  }
  // nextDomainTopologySpread / nextDomainAffinity / nextDomainAntiAffinity for a group on a dictionary key. Writes the
  // chosen domain set into sc.tq (words of the key). false = no domain satisfies the constraint.
  KS_DEV bool topo_next_domain(int g, bool self, const ReqRef& pod, const ReqRef& node) {
    const TopoView& T = P.topo;
    const Dict& d = P.dict;
    const int key = tt_key()[g], type = tt_type()[g];
    const uint32_t w0 = d.key_word_off[key], nw = d.key_word_off[key + 1] - w0;
    const uint64_t* D = tg_D() + (size_t)g * T.dom_words;
    const int32_t* cnt = tg_C() + (size_t)g * T.dom_words * 64;
    const uint16_t* rank = T.value_rank + (size_t)w0 * 64;
    uint64_t* tq = sc.tq;
    uint64_t any = 0;
    if (type == 0) {
      // domainMinCount — topologygroup.go:300-322
      long long mn = INT32_MAX;
      int supported = 0;
      for (uint32_t x = 0; x < nw; ++x) {
        const uint64_t dw = D[x];
        const uint64_t sup = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && topo_has(pod, key, w0 + x, b); });
        supported += popc64(sup);
        const uint64_t m = W::reduce_min(64, [&](int b) -> uint64_t { return ((sup >> b) & 1) ? (uint64_t)(uint32_t)cnt[x * 64 + b] : NONE; });
        if (m != NONE && (long long)m < mn) mn = (long long)m;
      }
      if (tt_min_domains()[g] >= 0 && supported < tt_min_domains()[g]) mn = 0;
      // the valid domain with the fewest pods, the smallest name among equals — topologygroup.go:251-297
      const long long skew = tt_max_skew()[g];
      uint64_t best = NONE;
      for (uint32_t x = 0; x < nw; ++x) {
        const uint64_t dw = D[x];
        const uint64_t m = W::reduce_min(64, [&](int b) -> uint64_t {
          if (!((dw >> b) & 1) || !topo_has(node, key, w0 + x, b)) return NONE;
          const long long c = (long long)cnt[x * 64 + b] + (self ? 1 : 0);
          if (c - mn > skew) return NONE;
          return ((uint64_t)c << 32) | ((uint64_t)rank[x * 64 + b] << 16) | (uint64_t)(x * 64 + b);
        });
        best = m < best ? m : best;
      }
      if (best == NONE) return false;
      const uint32_t v = (uint32_t)(best & 0xFFFF);
      W::for_n((int)nw, [&](int x) { tq[w0 + x] = ((uint32_t)x == (v >> 6)) ? (1ull << (v & 63)) : 0ull; });
      return true;
    }
    if (type == 2) {
      // empty domains the node and the pod admit — topologygroup.go:417-438
      for (uint32_t x = 0; x < nw; ++x) {
        const uint64_t dw = D[x];
        const uint64_t opt = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && cnt[x * 64 + b] == 0 && topo_has(node, key, w0 + x, b) && topo_has(pod, key, w0 + x, b); });
        W::store(&tq[w0 + x], opt);
        any |= opt;
      }
      W::sync();
      return any != 0;
    }
    // affinity — topologygroup.go:345-388
    uint64_t any_pod_nonzero = 0;
    for (uint32_t x = 0; x < nw; ++x) {
      const uint64_t dw = D[x];
      const uint64_t pn = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && cnt[x * 64 + b] > 0 && topo_has(pod, key, w0 + x, b); });
      const uint64_t opt = pn & W::ballot([&](int b) { return topo_has(node, key, w0 + x, b); });
      W::store(&tq[w0 + x], opt);
      any |= opt;
      any_pod_nonzero |= pn;
    }
    W::sync();
    if (any) return true;
    if (!self || !(tg_N()[g] == 0 || any_pod_nonzero == 0)) return false;
    // nothing to be affine to yet and the pod matches its own selector: bootstrap a domain (:372-386)
    uint64_t b1 = NONE, b2 = NONE;
    for (uint32_t x = 0; x < nw; ++x) {
      const uint64_t dw = D[x];
      const uint64_t ph = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && topo_has(pod, key, w0 + x, b); });
      const uint64_t nh = ph & W::ballot([&](int b) { return topo_has(node, key, w0 + x, b); });
      const uint64_t m1 = W::reduce_min(64, [&](int b) -> uint64_t { return ((nh >> b) & 1) ? (((uint64_t)rank[x * 64 + b] << 16) | (uint64_t)(x * 64 + b)) : NONE; });
      const uint64_t m2 = W::reduce_min(64, [&](int b) -> uint64_t { return ((ph >> b) & 1) ? (((uint64_t)rank[x * 64 + b] << 16) | (uint64_t)(x * 64 + b)) : NONE; });
      b1 = m1 < b1 ? m1 : b1;
      b2 = m2 < b2 ? m2 : b2;
    }
    if (b1 == NONE && b2 == NONE) return false;
    W::for_n((int)nw, [&](int x) {
      uint64_t v = 0;
      if (b1 != NONE && ((uint32_t)(b1 & 0xFFFF) >> 6) == (uint32_t)x) v |= 1ull << (b1 & 63);
      if (b2 != NONE && ((uint32_t)(b2 & 0xFFFF) >> 6) == (uint32_t)x) v |= 1ull << (b2 & 63);
      tq[w0 + x] = v;
    });
    return true;
  }
  // Domain values of group g's key for which the group's own nextDomain* would find a domain on a bin that admits the
  // value — as far as that can be said without looking at the bin (~0 = cannot tell). Used only to skip claims that
  // certainly fail; the winner still goes through topo_apply.
  KS_DEV uint64_t topo_ok_mask(int g, bool self, const ReqRef& pod) {
    const TopoView& T = P.topo;
    const Dict& d = P.dict;
    const int key = tt_key()[g], type = tt_type()[g];
    const uint32_t w0 = d.key_word_off[key];
    const uint64_t dw = tg_D()[(size_t)g * T.dom_words];
    const int32_t* cnt = tg_C() + (size_t)g * T.dom_words * 64;
    // the pod's own requirement on the key narrows the bin's domains (nodeRequirements = bin ∧ pod, nodeclaim.go:137-140)
    uint64_t podreq = ~0ull;
    {
      const uint32_t kdef = lo32(sc.cls[lay.k_f0()]), kc = hi32(sc.cls[lay.k_f0()]);
      const uint32_t kb = lo32(sc.cls[lay.k_f1()]) | hi32(sc.cls[lay.k_f1()]);
      if ((kb >> key) & 1) return ~0ull;
      if ((kdef >> key) & 1) podreq = ((kc >> key) & 1) ? (~sc.cls[lay.k_mask() + w0] & d.value_valid[w0]) : sc.cls[lay.k_mask() + w0];
    }
    if (type == 0) {
      const uint64_t sup = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && topo_has(pod, key, w0, b); });
      long long mn = INT32_MAX;
      const uint64_t m = W::reduce_min(64, [&](int b) -> uint64_t { return ((sup >> b) & 1) ? (uint64_t)(uint32_t)cnt[b] : NONE; });
      if (m != NONE) mn = (long long)m;
      if (tt_min_domains()[g] >= 0 && popc64(sup) < tt_min_domains()[g]) mn = 0;
      const long long skew = tt_max_skew()[g];
      const uint64_t valid = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && (long long)cnt[b] + (self ? 1 : 0) - mn <= skew; });
      return valid & podreq;
    }
    const uint64_t ph = W::ballot([&](int b) { return ((dw >> b) & 1) != 0 && topo_has(pod, key, w0, b); });
    if (type == 2) return ph & podreq & W::ballot([&](int b) { return cnt[b] == 0; });
    const uint64_t pn = ph & W::ballot([&](int b) { return cnt[b] > 0; });
    if (pn == 0 && self) return ~0ull;   // bootstrap path of nextDomainAffinity (:372-386): decided per bin
    return pn & podreq;
  }
  // Topology.AddRequirements + the Compatible check that follows it (topology.go:226-250, nodeclaim.go:195-206,
  // existingnode.go:111-122). `base` = the bin's requirements already intersected with the pod's. On success sc.topo
  // holds base ∧ every matching group's next domain and *changed tells whether that differs from base.
  KS_DEV bool topo_apply(const ReqRef& base, int bin_kind, int bin, bool allow_undefined, bool* changed) {
    const TopoView& T = P.topo;
    const Dict& d = P.dict;
    const ReqRef pod = P.cls_strict.at(d, (uint32_t)cur_class);
    ReqBuf& tb = sc.topo;
    W::for_n(d.req_words, [&](int w) { tb.mask[w] = base.mask[w]; });
    W::for_n(d.n_keys, [&](int k) {
      tb.gte[k] = (base.gte && bit(base.has_gte, k)) ? base.gte[k] : 0;
      tb.lte[k] = (base.lte && bit(base.has_lte, k)) ? base.lte[k] : 0;
      tb.minv[k] = base.minv ? base.minv[k] : -1;
    });
    uint32_t hm = 0;
    if (base.minv) for (int k = 0; k < d.n_keys; ++k) if (base.minv[k] >= 0) hm |= 1u << k;
    if (W::leader()) { tb.defined = base.defined; tb.complement = base.complement; tb.has_gte = base.has_gte; tb.has_lte = base.has_lte; tb.has_minv = hm; }
    W::sync();
    bool any_change = false;
    for (int tw = 0; tw < T.words; ++tw) for (uint64_t m = sc.t_match[tw]; m; m &= m - 1) {
      const int g = tw * 64 + ctz64(m);
      const bool self = (sc.t_sel[tw] >> (g & 63)) & 1;
      const int key = tt_key()[g];
      if (key < 0) {
        if (!topo_hostname_ok(g, self, pod, bin_kind, bin)) return false;
        continue;   // In [bin's hostname] ∧ the bin's own hostname requirement: nothing changes
      }
      if (!topo_next_domain(g, self, pod, base)) return false;
      ReqRef q;
      q.mask = sc.tq; q.defined = 1u << key; q.complement = 0; q.has_gte = q.has_lte = 0; q.gte = q.lte = nullptr; q.minv = nullptr;
      any_change = reqbuf_add(d, tb, q) || any_change;
      W::sync();
    }
    *changed = any_change;
    if (any_change) {
      ReqRef tr = tb.ref();
      if (reqs_compatible(d, base, tr, allow_undefined) != COMPAT_OK) return false;
    }
    return true;
  }
  // TopologyNodeFilter.Matches(taints, requirements) — topologynodefilter.go:68-96 (strict Compatible, :71)
  KS_DEV bool topo_filter_matches(int g, uint64_t taints, const ReqRef& fin, int bin_kind) {
    const TopoView& T = P.topo;
    const Dict& d = P.dict;
    if (tt_f_taint()[g] && (taints & ~tt_f_tolerates()[g])) return false;
    if (!tt_f_affinity()[g]) return true;
    const uint32_t a = tt_f_first()[g], b = tt_f_first()[g + 1];
    if (a == b) return true;
    const int hn = d.key_hostname;
    for (uint32_t i = a; i < b; ++i) {
      ReqRef r = T.f_reqs.at(d, i);
      if (bin_kind == 0 && hn >= 0 && bit(r.defined, hn)) {
        // a claim's hostname requirement (In [placeholder]) is kept implicit: only an open complement intersects it
        if (!(bit(r.complement, hn) && !bit(r.has_gte, hn) && !bit(r.has_lte, hn))) continue;
        r.defined &= ~(1u << hn);
      }
      if (reqs_compatible(d, fin, r, false) == COMPAT_OK) return true;
    }
    return false;
  }
  // Topology.Record — topology.go:197-220
  KS_DEV void topo_record(uint64_t taints, const ReqRef& fin, int bin_kind, int bin) {
    const TopoView& T = P.topo;
    const Dict& d = P.dict;
    for (int tw = 0; tw < T.words; ++tw)
    for (uint64_t todo = (sc.t_active[tw] & ~T.inverse_mask[tw] & sc.t_sel[tw]) | (T.inverse_mask[tw] & sc.t_owned[tw]); todo; todo &= todo - 1) {
      const int g = tw * 64 + ctz64(todo);
      const bool inv = (T.inverse_mask[tw] >> (g & 63)) & 1;
      if (!inv && !topo_filter_matches(g, taints, fin, bin_kind)) continue;
      const int key = tt_key()[g];
      if (key < 0) {
        const int32_t c = host_count_add(g, bin_kind, bin);
        if (c == 0) W::store(&tg_N()[g], tg_N()[g] + 1);
        if (bin_kind == 0 && tt_type()[g] != 1) {
          // threshold bitmaps of the scan prefilter: "count <= t-1" ends when the count reaches t, "count <= t" at t+1
          const long long t = tt_type()[g] == 2 ? 0 : (long long)tt_max_skew()[g];
          const long long n = (long long)c + 1;
          if (n == t || n == t + 1) {
            uint64_t* hz = S.host_le + ((size_t)tt_host_slot()[g] * 2 + (n == t ? 0 : 1)) * S.claim_words + (bin >> 6);
            W::store(hz, (uint64_t)(*hz & ~(1ull << (bin & 63))));
          }
        }
        W::sync();
        continue;
      }
      if (!bit(fin.defined, key)) continue;   // Exists: no values
      const uint32_t w0 = d.key_word_off[key], nw = d.key_word_off[key + 1] - w0;
      const bool anti = inv || tt_type()[g] == 2;
      int nvals = 0;
      for (uint32_t x = 0; x < nw; ++x) nvals += popc64(fin.mask[w0 + x]);
      // anti-affinity blocks every stored value (domains.Values(), also for a NotIn set); the others count a pod only
      // once its domain is decided (topology.go:203-211)
      if (!anti && (bit(fin.complement, key) || nvals != 1)) continue;
      uint64_t* D = tg_D() + (size_t)g * T.dom_words;
      int32_t* cnt = tg_C() + (size_t)g * T.dom_words * 64;
      int fresh = 0;
      for (uint32_t x = 0; x < nw; ++x) {
        const uint64_t mk = fin.mask[w0 + x];
        if (!mk) continue;
        const uint64_t was_zero = W::ballot([&](int b) {
          if (!((mk >> b) & 1)) return false;
          const int32_t c = cnt[x * 64 + b];
          cnt[x * 64 + b] = c + 1;
          return c == 0;
        });
        fresh += popc64(was_zero);
        W::store(&D[x], (uint64_t)(D[x] | mk));
      }
      if (fresh) W::store(&tg_N()[g], tg_N()[g] + fresh);
      W::sync();
    }
  }
  KS_DEV ReqRef out_ref(const uint64_t* cold) const {   // the record being committed (sc.out) as a requirement set
    ReqRef r;
    const uint64_t f0 = sc.out[lay.c_f0()], f1 = sc.out[lay.c_f1()];
    r.mask = sc.out + lay.c_mask(); r.defined = lo32(f0); r.complement = hi32(f0); r.has_gte = lo32(f1); r.has_lte = hi32(f1);
    r.gte = (const int64_t*)cold; r.lte = (const int64_t*)(cold + lay.nk); r.minv = nullptr;
    return r;
  }

  // NodeClaim.CanAdd (nodeclaim.go:124-242) for a pod of the class in sc.cls on the bin record `bin`/`bin_cold`.
  // On success sc.out holds the committed record's masks and flag words, sc.its / sc.total the new instance types and
  // requests (finish_record completes head/meta and writes the record).
  // Number of distinct values the instance types in `its` carry on `key` (InstanceTypes.SatisfiesMinValues,
  // types.go:399-433: the union of requirement.Values() over the types). One lane per dictionary value; a value counts
  // when some surviving type lists it (per-value instance-type bitmasks from the it_index kernel).
  KS_DEV int distinct_values(int key, const uint64_t* its) {
    const Dict& d = P.dict;
    const LdsTables& Lt = L;
    const int iw = lay.iw, nk = d.n_keys;
    int total = 0;
    if (key == d.key_it) {   // every type requires In [own name]
      for (int w = 0; w < iw; ++w) total += popc64(its[w]);
      return total;
    }
    for (uint32_t x = d.key_word_off[key]; x < d.key_word_off[key + 1]; ++x) {
      const uint64_t vb = d.value_valid[x];
      total += popc64(W::ballot([&](int b) {
        if (!((vb >> b) & 1)) return false;
        const uint16_t sl = Lt.kvslot[x * 64 + b];
        // kv = the types whose requirement HAS the value. A type lists it when it is an In requirement that has it, or a
        // complement (NotIn) that does not: Values() of a NotIn requirement are the excluded ones, of Exists none.
        uint64_t any = 0;
        for (int w = 0; w < iw; ++w) {
          const uint64_t has = sl == 0xFFFF ? 0ull : Lt.kv[(size_t)sl * iw + w], cm = Lt.keymask[(size_t)(nk + key) * iw + w];
          any |= ((has & ~cm) | (~has & cm)) & its[w];
        }
        return any != 0;
      }));
    }
    return total;
  }
  // minValues of the requirement set (mv, one per key, -1 = nil) against the surviving instance types (sc.its):
  // nodeclaim.go:602-613. Strict policy: unmet => false. BestEffort (relax): the requirement is lowered to what the types
  // offer (nodeclaim.go:224-229) and *lowered is set.
  KS_DEV bool min_values_ok(int32_t* mv, bool relax, bool* lowered) {
    bool ok = true;
    for (int k = 0; k < lay.nk; ++k) {
      const int32_t want = mv[k];
      if (want < 0) continue;
      const int have = distinct_values(k, sc.its);
      if (have >= want) continue;
      if (!relax) { ok = false; continue; }
      W::store(&mv[k], (int32_t)have);
      *lowered = true;
    }
    W::sync();
    return ok;
  }
  // Reservation ids of the available reserved offerings of the surviving instance types (sc.its) that are compatible
  // with the requirements (reqs.IsCompatible(offering.Requirements, AllowUndefinedWellKnownLabels), nodeclaim.go:318-321).
  // One lane per instance type.
  KS_DEV uint64_t reservable_ids(const ReqRef& r) {
    const Dict& d = P.dict;
    const ProblemView& Pv = P;
    if (Pv.ct_reserved < 0 || Pv.key_rid < 0) return 0;
    if (d.key_ct >= 0 && bit(r.defined, d.key_ct) && !req_has(d, r, d.key_ct, d.key_word_off[d.key_ct], Pv.ct_reserved)) return 0;
    uint32_t zones = 0;
    if (d.key_zone >= 0 && bit(r.defined, d.key_zone)) { for (int z = 0; z < P.n_zones; ++z) if (req_has(d, r, d.key_zone, d.key_word_off[d.key_zone], z)) zones |= 1u << z; }
    else zones = (1u << P.n_zones) - 1;
    uint64_t ids_ok = 0;   // reservation ids the requirement on the reservation-id key admits
    const bool rid_defined = bit(r.defined, Pv.key_rid);
    if (!rid_defined) { if (bit(d.well_known_mask, Pv.key_rid)) ids_ok = ~0ull; }
    else for (int i = 0; i < Pv.n_resv; ++i) if (req_has(d, r, Pv.key_rid, d.key_word_off[Pv.key_rid] + (uint32_t)(i >> 6), i & 63)) ids_ok |= 1ull << i;
    if (!ids_ok) return 0;
    const uint64_t* sits = sc.its;
    return W::reduce_or(P.n_its, [&](int it) -> uint64_t {
      if (!((sits[it >> 6] >> (it & 63)) & 1)) return 0;
      uint64_t m = 0;
      for (uint32_t o = Pv.it_resv_first[it]; o < Pv.it_resv_first[it + 1]; ++o) if ((zones >> Pv.resv_zone[o]) & 1) m |= 1ull << Pv.resv_id[o];
      return m & ids_ok;
    });
  }
  // NodeClaim.Add's reservation bookkeeping (nodeclaim.go:255-262): reserve what is new, release what is no longer held
  KS_DEV void commit_reservations(int c, bool fresh) {
    const uint64_t held = fresh ? 0ull : S.c_reserved[c];
    const uint64_t now = pending_reserved;
    if (W::leader()) {
      for (uint64_t m = now & ~held; m; m &= m - 1) sc.resv_cap[ctz64(m)] -= 1;
      for (uint64_t m = held & ~now; m; m &= m - 1) sc.resv_cap[ctz64(m)] += 1;
    }
    W::store(&S.c_reserved[c], now);
    W::sync();
  }
  // dst <- src (wave-cooperative)
  KS_DEV void reqbuf_copy(ReqBuf& dst, const ReqBuf& src) {
    W::for_n(lay.rw, [&](int w) { dst.mask[w] = src.mask[w]; });
    W::for_n(lay.nk, [&](int k) { dst.gte[k] = src.gte[k]; dst.lte[k] = src.lte[k]; dst.minv[k] = src.minv[k]; });
    if (W::leader()) { dst.defined = src.defined; dst.complement = src.complement; dst.has_gte = src.has_gte; dst.has_lte = src.has_lte; dst.has_minv = src.has_minv; }
    W::sync();
  }
  // dst <- the requirement set r (reqbuf_load, wave-cooperative)
  KS_DEV void reqbuf_copy_in(ReqBuf& dst, const ReqRef& r) {
    W::for_n(lay.rw, [&](int w) { dst.mask[w] = r.mask[w]; });
    uint32_t hm = 0;
    if (r.minv) for (int k = 0; k < lay.nk; ++k) if (r.minv[k] >= 0) hm |= 1u << k;
    W::for_n(lay.nk, [&](int k) {
      dst.gte[k] = (r.gte && bit(r.has_gte, k)) ? r.gte[k] : 0;
      dst.lte[k] = (r.lte && bit(r.has_lte, k)) ? r.lte[k] : 0;
      dst.minv[k] = r.minv ? r.minv[k] : -1;
    });
    if (W::leader()) { dst.defined = r.defined; dst.complement = r.complement; dst.has_gte = r.has_gte; dst.has_lte = r.has_lte; dst.has_minv = hm; }
    W::sync();
  }
  // the merged requirement set `m` becomes the record being built (sc.out / sc.out_cold)
  KS_DEV ReqRef reqbuf_to_out(const ReqBuf& m) {
    uint64_t* o = sc.out;
    W::for_n(lay.rw, [&](int w) { o[w] = m.mask[w]; });
    if (W::leader()) {
      o[lay.c_f0()] = (uint64_t)m.defined | ((uint64_t)m.complement << 32);
      o[lay.c_f1()] = (uint64_t)m.has_gte | ((uint64_t)m.has_lte << 32);
    }
    int64_t* cg = (int64_t*)sc.out_cold; int64_t* cl = cg + lay.nk; int32_t* cv = (int32_t*)(sc.out_cold + 2 * lay.nk);
    W::for_n(lay.nk, [&](int k) { cg[k] = m.gte[k]; cl[k] = m.lte[k]; cv[k] = m.minv[k]; });
    ReqRef r;
    r.mask = sc.out + lay.c_mask(); r.defined = m.defined; r.complement = m.complement;
    r.has_gte = m.has_gte; r.has_lte = m.has_lte; r.gte = cg; r.lte = cl; r.minv = cv;
    return r;
  }
  KS_DEV int can_add(const uint64_t* bin, const uint64_t* bin_cold, bool fresh, bool want_diag, bool* reqs_changed, bool* its_changed, int claim_id) {
    const Dict& d = P.dict;
    const int nr = lay.nr;
    cadd(ctr.bin_evaluations, 1);
    topo_reached = false;
    if (FULL && P.hp_on) bin_hp = claim_id >= 0 ? S.c_hp[claim_id] : 0ull;
    unsigned long long ta = W::clock();
    const uint64_t bin_taints = sc.tmpl_taints[lo32(bin[lay.c_meta()]) & 31u];
    if (bin_taints & ~sc.cls[lay.k_tol()]) return E_TAINTS;                              // Taints.ToleratesPod — nodeclaim.go:126
    const int64_t* req = (const int64_t*)(sc.cls + lay.k_req());
    const int64_t* head = (const int64_t*)(bin + lay.c_head());
    const int64_t* tot = (const int64_t*)(bin + lay.c_total());
    if (W::ballot([&](int l) { return l < nr && req[l] > head[l]; })) return E_INSTANCE_TYPES;   // no remaining instance type can hold it
    const bool bin_minv = FULL && (hi32(bin[lay.c_meta2()]) & 2u) != 0, cls_minv = FULL && (lo32(sc.cls[lay.k_meta()]) & 1u) != 0;
    const bool slow = FULL && (bin[lay.c_f1()] != 0 || sc.cls[lay.k_f1()] != 0 || cls_minv || lay.rw > 64);   // lite: no bounds anywhere, rw <= 64
    uint32_t kdef = lo32(sc.cls[lay.k_f0()]);
    int hn = d.key_hostname;
    if (hn >= 0 && ((kdef >> hn) & 1)) {
      // the claim's own hostname requirement is In [hostname-placeholder-N] (nodeclaim.go:97), a value outside every
      // dictionary: only an unbounded complement (NotIn / Exists) on the pod side intersects it.
      uint32_t kc = hi32(sc.cls[lay.k_f0()]);
      uint64_t kf1 = sc.cls[lay.k_f1()];
      if (!((kc >> hn) & 1) || (((lo32(kf1) | hi32(kf1)) >> hn) & 1)) return E_INCOMPATIBLE;
      kdef &= ~(1u << hn);
    }
    bool changed = false;
    ReqRef merged;
    unsigned long long tb = W::clock();
    ctr.cycles[11] += tb - ta;
    if (!slow) {
      if (!fast_compat_merge(bin, kdef, &changed)) return E_INCOMPATIBLE;               // nodeclaim.go:133-136
      merged.mask = sc.out + lay.c_mask(); merged.defined = lo32(sc.out[lay.c_f0()]); merged.complement = hi32(sc.out[lay.c_f0()]);
      merged.has_gte = merged.has_lte = 0; merged.gte = merged.lte = nullptr; merged.minv = nullptr;
      if (bin_minv) {  // minValues of the bin carry over (pods cannot add any)
        const uint64_t* bc = bin_cold; uint64_t* oc = sc.out_cold;
        W::for_n(lay.cold_words(), [&](int i) { oc[i] = bc[i]; });
        merged.minv = (const int32_t*)(sc.out_cold + 2 * lay.nk);   // a topology step that rewrites the set must keep them
      }
    } else {
      ReqRef br = claim_ref(bin, bin_cold);
      ReqRef q = class_ref(sc.cls, sc.cls_cold);
      q.defined = kdef;
      if (reqs_compatible(d, br, q, true) != COMPAT_OK) return E_INCOMPATIBLE;
      reqbuf_load(d, sc.merged, br);
      changed = reqbuf_add(d, sc.merged, q);
      merged = reqbuf_to_out(sc.merged);
    }
    if (FULL && cur_vol_n) {
      // volume requirement alternatives — nodeclaim.go:138-157: each starts from bin ∧ pod, the first one that passes the rest
      // of CanAdd wins, the error kept is the last one's
      reqbuf_copy_in(sc.vbase, merged);
      int err = E_INCOMPATIBLE;
      for (uint32_t va = 0; va < cur_vol_n; ++va) {
        ReqRef vr = P.vol_reqs.at(d, cur_vol_first + va);
        if (hn >= 0 && bit(vr.defined, hn)) {
          // against the claim's own hostname In [hostname-placeholder-N]: only an unbounded complement intersects it
          if (!bit(vr.complement, hn) || bit(vr.has_gte | vr.has_lte, hn)) { err = E_INCOMPATIBLE; continue; }
          vr.defined &= ~(1u << hn);
        }
        if (reqs_compatible(d, sc.vbase.ref(), vr, true) != COMPAT_OK) { err = E_INCOMPATIBLE; continue; }   // nodeclaim.go:170-173
        reqbuf_copy(sc.merged, sc.vbase);
        const bool vch = reqbuf_add(d, sc.merged, vr);
        W::sync();
        err = can_add_rest(bin, fresh, want_diag, reqs_changed, its_changed, claim_id, reqbuf_to_out(sc.merged), changed || vch, bin_minv);
        if (err == E_OK) return E_OK;
      }
      return err;
    }
    return can_add_rest(bin, fresh, want_diag, reqs_changed, its_changed, claim_id, merged, changed, bin_minv);
  }
  // tryVolumeAlternative past the volume requirements (nodeclaim.go:195-242): topology, requests, the instance-type filter,
  // minValues, the reservations. `merged` = the requirement set so far, already in the record being built (sc.out / sc.out_cold).
  KS_DEV int can_add_rest(const uint64_t* bin, bool fresh, bool want_diag, bool* reqs_changed, bool* its_changed, int claim_id, ReqRef merged, bool changed, bool bin_minv) {
    const int nr = lay.nr;
    const int64_t* req = (const int64_t*)(sc.cls + lay.k_req());
    const int64_t* tot = (const int64_t*)(bin + lay.c_total());
    unsigned long long tb = W::clock();
    if (FULL && cur_M) {
      // topology: nodeclaim.go:195-208
      topo_reached = true;
      bool tchanged = false;
      if (!topo_apply(merged, 0, claim_id, true, &tchanged)) return E_TOPOLOGY;
      if (tchanged) { merged = reqbuf_to_out(sc.topo); changed = true; }
    }
    KS_DIAG(cadd(ctr.full_evaluations, 1));
    unsigned long long tc = W::clock();
    ctr.cycles[12] += tc - tb;
    if (reqs_changed) *reqs_changed = changed;
    int64_t* ntot = sc.total;
    W::for_n(nr, [&](int r) { ntot[r] = tot[r] + req[r]; });                               // resources.Merge — nodeclaim.go:211
    const bool full = changed || fresh;
    unsigned long long td = W::clock();
    ctr.cycles[13] += td - tc;
    const bool any_it = filter_instance_types(bin + lay.c_its(), sc.total, full, merged, want_diag, (int)(lo32(bin[lay.c_meta()]) & 31u));
    ctr.cycles[14] += W::clock() - td;
    if (!any_it) return E_INSTANCE_TYPES;  // nodeclaim.go:213
    minv_lowered = false;
    if (bin_minv) {
      // SatisfiesMinValues over the remaining types (nodeclaim.go:602-613); only a new NodeClaim may relax (scheduler.go:731)
      bool lowered = false;
      if (!min_values_ok((int32_t*)(sc.out_cold + 2 * lay.nk), fresh && min_values_best_effort, &lowered)) { last_diag |= 64; return E_MIN_VALUES; }
      if (lowered) { minv_lowered = true; if (reqs_changed) *reqs_changed = true; }
    }
    if (FULL && P.reserved_on) {
      // offeringsToReserve — nodeclaim.go:303-350
      const uint64_t held = claim_id >= 0 ? S.c_reserved[claim_id] : 0ull;
      const uint64_t cand = reservable_ids(merged);
      uint64_t open_ids = 0;
      for (int i = 0; i < P.n_resv; ++i) if (sc.resv_cap[i] > 0) open_ids |= 1ull << i;
      const uint64_t out_ids = cand & (held | open_ids);               // ReservationManager.CanReserve (reservationmanager.go:62-72)
      if (P.reserved_strict && out_ids == 0 && (cand != 0 || held != 0)) { topo_reached = true; return E_RESERVED; }
      pending_reserved = out_ids;
    }
    if (its_changed) {
      const uint64_t* bi = bin + lay.c_its();
      const uint64_t* ni = sc.its;
      const int iw = lay.iw;
      *its_changed = W::ballot([&](int l) { return l < iw && bi[l] != ni[l]; }) != 0;
    }
    return E_OK;
  }

  // completes sc.out (its, total, head, meta words) and writes the record to HBM (and the record cache)
  KS_DEV void finish_record(int c, const uint64_t* bin, bool recompute_head, uint32_t tmpl, uint32_t npods, uint32_t seq, uint32_t flags_hi, bool write_cold) {
    const int nr = lay.nr, iw = lay.iw, np = iw * 64;
    uint64_t* o = sc.out;
    const uint64_t* sits = sc.its;
    const RecLayout ly = lay;
    const LdsTables& Lt = L;
    const int64_t* ntot = sc.total;
    const int hg0 = FULL ? sc.dg_first[tmpl & 31u] : 0, hg1 = FULL ? sc.dg_first[(tmpl & 31u) + 1] : 1;
    if (FULL && recompute_head && hg1 - hg0 > 1) {
      // several daemon-overhead groups: headroom = max over groups of (max allocatable in the group - its overhead) - total
      for (int r = 0; r < nr; ++r) {
        int64_t best = INT64_MIN;
        for (int g = hg0; g < hg1; ++g) {
          const uint64_t* gm = Lt.dg_its + (size_t)g * iw;
          int64_t mx = W::reduce_max_i64(np, [&](int it) { return ((sits[it >> 6] & gm[it >> 6]) >> (it & 63)) & 1 ? Lt.alloc[(size_t)r * np + it] : INT64_MIN; });
          if (mx != INT64_MIN && mx - Lt.dg_ov[(size_t)g * nr + r] > best) best = mx - Lt.dg_ov[(size_t)g * nr + r];
        }
        if (W::leader()) o[ly.c_head() + r] = (uint64_t)(best - ntot[r] + P.xg_bonus[r]);
      }
    } else if (SC::kRegTables && recompute_head && regs_ok) {
      // headroom = max allocatable over the surviving instance types - total, from the register tables
      uint64_t sw[kRegIw];
#pragma unroll
      for (int j = 0; j < kRegIw; ++j) sw[j] = j < iw ? sits[j] : 0ull;
#pragma unroll
      for (int r = 0; r < kRegNr; ++r) {
        if (r >= nr) break;
        int64_t mx = W::lanes_max_i64([&](int l) {
          int64_t m = INT64_MIN;
#pragma unroll
          for (int j = 0; j < kRegIw; ++j) { const int64_t a = ((sw[j] >> l) & 1) ? RA(l, j, r) : INT64_MIN; m = a > m ? a : m; }
          return m;
        });
        if (W::leader()) o[ly.c_head() + r] = (uint64_t)(mx - (FULL ? Lt.dg_ov[(size_t)hg0 * nr + r] : 0) - ntot[r]);
      }
    } else if (recompute_head) {
      // headroom = max allocatable over the surviving instance types - total
      for (int r = 0; r < nr; ++r) {
        int64_t mx = W::reduce_max_i64(np, [&](int it) { return ((sits[it >> 6] >> (it & 63)) & 1) ? Lt.alloc[(size_t)r * np + it] : INT64_MIN; });
        if (W::leader()) o[ly.c_head() + r] = (uint64_t)(mx - (FULL ? Lt.dg_ov[(size_t)hg0 * nr + r] : 0) - ntot[r] + (FULL ? P.xg_bonus[r] : 0));   // an override group may hold more than the base allocatable
      }
    } else {
      const int64_t* bh = (const int64_t*)(bin + ly.c_head());
      const int64_t* bt = (const int64_t*)(bin + ly.c_total());
      W::for_n(nr, [&](int r) { o[ly.c_head() + r] = (uint64_t)(bh[r] - (ntot[r] - bt[r])); });  // same types, same maximum
    }
    W::for_n(iw > nr ? iw : nr, [&](int i) {
      if (i < iw) o[ly.c_its() + i] = sits[i];
      if (i < nr) o[ly.c_total() + i] = (uint64_t)ntot[i];
    });
    if (W::leader()) {
      o[ly.c_meta()] = (uint64_t)tmpl | ((uint64_t)npods << 32);
      o[ly.c_meta2()] = (uint64_t)seq | ((uint64_t)flags_hi << 32);
    }
    W::sync();
    const int64_t* mr = sc.min_request;
    const int64_t* oh = (const int64_t*)(o + ly.c_head());
    const bool is_closed = W::ballot([&](int l) { return l < nr && mr[l] > 0 && oh[l] < mr[l]; }) != 0;
    uint64_t* dst = S.c_hot + (size_t)c * ly.c_hot_words();
    uint64_t* line = L.cache + (size_t)(c & kLineMask) * ly.c_hot_words();
    W::for_n(ly.c_hot_words(), [&](int i) { uint64_t v = o[i]; dst[i] = v; line[i] = v; });
    { int64_t* hd = S.c_headroom; const int mc = S.max_claims; W::for_n(nr, [&](int r) { hd[(size_t)r * mc + c] = oh[r]; }); }
    if (FULL && P.topo.n_key_slots) {
      // values the claim still admits on each topology key (undefined key or bounds: everything)
      const TopoView& T = P.topo;
      const Dict& d = P.dict;
      uint64_t* km = S.c_keymask;
      const int mc = S.max_claims;
      const uint32_t def = lo32(o[ly.c_f0()]), comp = hi32(o[ly.c_f0()]), bnd = lo32(o[ly.c_f1()]) | hi32(o[ly.c_f1()]);
      W::for_n(T.n_key_slots, [&](int sl) { sc.km_old[sl] = km[(size_t)sl * mc + c]; });
      W::for_n(T.n_key_slots, [&](int sl) {
        const int key = T.slot_key[sl];
        const uint32_t x = d.key_word_off[key];
        uint64_t v = ~0ull;
        if (((def >> key) & 1) && !((bnd >> key) & 1)) v = ((comp >> key) & 1) ? (~o[ly.c_mask() + x] & d.value_valid[x]) : o[ly.c_mask() + x];
        km[(size_t)sl * mc + c] = v;
      });
      // the inverse table: bit c of kv_claims[slot][value] follows the claim's admitted values (a new claim starts from none)
      uint64_t* kvc = S.kv_claims;
      const int cw = S.claim_words;
      for (int sl = 0; sl < T.n_key_slots; ++sl) {
        const uint32_t x = d.key_word_off[T.slot_key[sl]];
        const uint64_t now = km[(size_t)sl * mc + c] & d.value_valid[x];
        const uint64_t before = npods == 1 ? 0ull : (sc.km_old[sl] & d.value_valid[x]);
        const uint64_t diff = now ^ before;
        if (!diff) continue;
        const uint64_t bitc = 1ull << (c & 63);
        W::for_n(64, [&](int v) {
          if (!((diff >> v) & 1)) return;
          uint64_t* wp = kvc + ((size_t)sl * 64 + v) * cw + (c >> 6);
          *wp = ((now >> v) & 1) ? (*wp | bitc) : (*wp & ~bitc);
        });
      }
    }
    if (W::leader()) sc.cache_tag[c & kLineMask] = c;
    if (write_cold) {
      uint64_t* dc = S.c_cold + (size_t)c * ly.cold_words();
      const uint64_t* oc = sc.out_cold;
      W::for_n(ly.cold_words(), [&](int i) { dc[i] = oc[i]; });
    }
    if (is_closed && W::leader()) L.closed[c >> 6] |= 1ull << (c & 63);
    W::sync();
  }
  // Keys whose Operator() is Exists (requirement.go:290-301): a complement set that excludes nothing (bare Exists, Gt, Lt).
  // `mask` = the record's requirement words at stride `stride`.
  template <class M>
  KS_DEV uint32_t exists_keys(uint32_t complement, M mask_word) {
    if (!complement) return 0;
    const uint8_t* wk = sc.word_key;
    const uint32_t nonempty = (uint32_t)W::reduce_or(lay.rw, [&](int l) { return mask_word(l) ? (uint64_t)1 << wk[l] : (uint64_t)0; });
    return complement & ~nonempty;
  }
  // Does a commit that turned requirement set `before` into `after` void the rejections cached for this bin? A class
  // rejected before topology stays rejected while the bin only narrows (value sets, instance types and headroom shrink; bound
  // host ports grow). Two ways back, both through Requirements.Compatible:
  //   * a key becomes DEFINED on the bin: a pod's In / Exists on a custom label the bin did not carry is "undefined key"
  //     (requirements.go:185-193) until another pod's NotIn / DoesNotExist has put the key there;
  //   * a key's operator goes from Exists to NotIn (another pod's NotIn narrowed a bare Exists / Gt / Lt): Intersects lets a
  //     NotIn / DoesNotExist pair through whatever their values (requirements.go:258-265), so a `k DoesNotExist` pod the
  //     `k Exists` bin rejected is compatible with the `k NotIn [x]` bin.
  // Every other operator change leaves the NotIn / DoesNotExist class or stays inside it.
  KS_DEV bool revives_rejections(uint32_t def_before, uint32_t def_after, uint32_t exists_before, uint32_t exists_after) {
    return def_before != def_after || (exists_before & ~exists_after) != 0;
  }
  KS_DEV void reset_column(int c) {
    KS_DIAG(cadd(ctr.column_resets, 1));
    uint64_t* dead = S.dead;
    const int cw = S.claim_words;
    const uint64_t clr = ~(1ull << (c & 63));
    const int word = c >> 6;
    W::for_n(P.n_classes, [&](int k) { dead[(size_t)k * cw + word] &= clr; });
  }
  KS_DEV void commit_pod(int pod, int claim, uint32_t slot) {
    (void)pod;
    W::store(&S.assign[cur_out], (int32_t)claim);
    W::store(&S.slot[cur_out], slot);
  }

  // ---- in-flight scan: addToInflightNode (scheduler.go:658-692) ---------------------------------------------
  // One probe of claim c. On failure the caller records the verdict in the class's dead row — unless the probe got as
  // far as the topology stage: domain counters move with every commit anywhere, so that verdict cannot be cached.
  KS_DEV int try_claim(int k, int c, int pod) {
    unsigned long long t0 = W::clock();
    const RecLayout ly = lay;
    // hot record: from the LDS record cache when this claim was the last one committed to its line, else one coalesced load
    if (sc.cache_tag[c & kLineMask] == c) load_words(sc.claim, L.cache + (size_t)(c & kLineMask) * ly.c_hot_words(), ly.c_hot_words());
    else load_words(sc.claim, S.c_hot + (size_t)c * ly.c_hot_words(), ly.c_hot_words());
    const uint64_t f1 = sc.claim[ly.c_f1()];
    const uint32_t m2 = hi32(sc.claim[ly.c_meta2()]);
    if (FULL && (f1 != 0 || (m2 & 2u))) load_words(sc.claim_cold, S.c_cold + (size_t)c * ly.cold_words(), ly.cold_words());
    bool changed = false, its_changed = false;
    unsigned long long t1 = W::clock();
    ctr.cycles[4] += t1 - t0;
    int rc = can_add(sc.claim, sc.claim_cold, false, false, &changed, &its_changed, c);
    unsigned long long t2 = W::clock();
    ctr.cycles[5] += t2 - t1;
    if (rc != E_OK) return rc;
    const uint32_t tmpl = lo32(sc.claim[ly.c_meta()]), np = hi32(sc.claim[ly.c_meta()]);
    cadd(ctr.ref_bin_evaluations, (unsigned long long)((unsigned long long)order.position(c) + 1));   // the reference walked every claim up to this position
    if (!changed) {
      // requirements untouched: carry masks and flags over from the bin record
      uint64_t* o = sc.out;
      const uint64_t* b = sc.claim;
      W::for_n(ly.rw + 2, [&](int w) { int i = w < ly.rw ? w : ly.c_f0() + (w - ly.rw); o[i] = b[i]; });
    }
    const bool out_cold = FULL && changed && (sc.out[ly.c_f1()] != 0 || (m2 & 2u));
    if (FULL && cur_rec) topo_record(sc.tmpl_taints[tmpl & 31u], out_ref(changed ? sc.out_cold : sc.claim_cold), 0, c);   // nodeclaim.go:252-253
    if (FULL && P.reserved_on) commit_reservations(c, false);
    if (FULL && P.hp_on && cur_hp_use) W::store(&S.c_hp[c], (uint64_t)(S.c_hp[c] | cur_hp_use));   // HostPortUsage.Add — nodeclaim.go:256-259
    finish_record(c, sc.claim, its_changed, tmpl, np + 1, lo32(sc.claim[ly.c_meta2()]), m2, out_cold);
    order.increment(c);
    // the column of cached rejections is cleared only when the new requirement set can bring a class back (revives_rejections)
    if (changed) {
      const uint64_t* cb = sc.claim + ly.c_mask(); const uint64_t* co = sc.out + ly.c_mask();
      const uint32_t ex_b = exists_keys(hi32(sc.claim[ly.c_f0()]), [&](int l) { return cb[l]; });
      const uint32_t ex_a = ex_b ? exists_keys(hi32(sc.out[ly.c_f0()]), [&](int l) { return co[l]; }) : 0;
      if (revives_rejections(lo32(sc.claim[ly.c_f0()]), lo32(sc.out[ly.c_f0()]), ex_b, ex_a)) reset_column(c);
    }
    commit_pod(pod, c, np);
    ctr.cycles[6] += W::clock() - t2;
    return E_OK;
  }
  KS_DEV bool scan_inflight(int k, int pod) {
    if (n_claims == 0) return false;
    const int words = (n_claims + 63) >> 6;
    // n_claims <= LdsPlan::order_cap <= 8192, so the live set always fits two words per lane
    uint64_t* drow = S.dead + (size_t)k * S.claim_words;
    const uint64_t* closed = L.closed;
    const int nc = n_claims;
    uint64_t* stage = BIG ? L.stage_big : sc.stage;
    unsigned long long ts0 = W::clock();
    // One coalesced load of the class's dead row (lane l holds words l and l + 64: up to 8192 claims), live = not dead,
    // not closed, staged in LDS so that everything after it is LDS-only.
    uint64_t any;
    if constexpr (BIG) {
      any = W::ballot([&](int l) {
        uint64_t acc = 0;
        for (int w = l; w < words; w += 64) {
          const uint64_t valid = (w == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
          const uint64_t a = ~drow[w] & ~closed[w] & valid;
          stage[w] = a;
          acc |= a;
        }
        return acc != 0;
      });
    } else {
      any = W::ballot([&](int l) {
        uint64_t a = 0, b = 0;
        if (l < words) {
          uint64_t valid = (l == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
          a = ~drow[l] & ~closed[l] & valid;
        }
        if (l + 64 < words) {
          uint64_t valid = (l + 64 == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
          b = ~drow[l + 64] & ~closed[l + 64] & valid;
        }
        stage[l] = a; stage[l + 64] = b;
        return (a | b) != 0;
      });
    }
    W::sync();
    if (!any) return false;
    if (FULL && cur_M) {
      // Hostname spread / anti-affinity groups of the class: a claim whose per-claim counter already rules it out
      // (topologygroup.go:240-247,409-414) cannot pass CanAdd, whatever else holds. One lane per claim, counters coalesced.
      const TopoView& T = P.topo;
      for (int tw = 0; tw < T.words; ++tw) for (uint64_t m = sc.t_match[tw]; m; m &= m - 1) {
        const int g = tw * 64 + ctz64(m);
        const bool self = (sc.t_sel[tw] >> (g & 63)) & 1;
        if (tt_key()[g] >= 0) {
          // dictionary key: skip the claims whose admitted values miss every domain the group could pick
          const int sl = tt_key_slot()[g];
          if (sl < 0) continue;
          const uint64_t okv = topo_ok_mask(g, self, P.cls_strict.at(P.dict, (uint32_t)cur_class));
          if (okv == ~0ull) continue;
          // one lane per 64 claims: OR the "claims that admit value v" words of the eligible values
          const uint64_t* kvc = S.kv_claims + (size_t)sl * 64 * S.claim_words;
          const int cw = S.claim_words;
          const uint64_t vals = okv & P.dict.value_valid[P.dict.key_word_off[tt_key()[g]]];
          any = W::ballot([&](int l) {
            uint64_t acc_any = 0;
            for (int w = l; w < words; w += 64) {
              const uint64_t st = stage[w];
              if (!st) continue;
              uint64_t acc = 0;
              for (uint64_t vv = vals; vv; vv &= vv - 1) acc |= kvc[(size_t)ctz64(vv) * cw + w];
              const uint64_t v = st & acc;
              stage[w] = v;
              acc_any |= v;
            }
            return acc_any != 0;
          });
          W::sync();
          if (!any) return false;
          continue;
        }
        if (tt_type()[g] == 1) continue;
        const long long limit = tt_type()[g] == 2 ? 0 : (long long)tt_max_skew()[g] - (self ? 1 : 0);
        if (limit < 0) return false;   // not even an empty claim satisfies it
        {
          // "count <= limit" from the threshold bitmap of the group (limit is t-1 or t): one word per 64 claims
          const long long t = tt_type()[g] == 2 ? 0 : (long long)tt_max_skew()[g];
          const uint64_t* hz = S.host_le + ((size_t)tt_host_slot()[g] * 2 + (limit == t ? 1 : 0)) * S.claim_words;
          any = W::ballot([&](int l) {
            uint64_t acc_any = 0;
            for (int w = l; w < words; w += 64) { const uint64_t v = stage[w] & hz[w]; stage[w] = v; acc_any |= v; }
            return acc_any != 0;
          });
          W::sync();
          if (!any) return false;
        }
      }
    }
    unsigned long long ts1 = W::clock();
    ctr.cycles[17] += ts1 - ts0;
    // CanAdd's first resource test — a request larger than what the largest surviving instance type still has free
    // (nodeclaim.go:213 can only fail) — is folded into the selection below from the SoA headroom table: these failures
    // are exact and permanent until the claim's column resets.
    const int64_t* hd = S.c_headroom;
    const int mc = S.max_claims;
    const int nr = lay.nr;
    const int64_t* req = (const int64_t*)(sc.cls + lay.k_req());
    int64_t rq[kRegNr];
#pragma unroll
    for (int r = 0; r < kRegNr; ++r) rq[r] = r < nr ? req[r] : INT64_MIN;
    unsigned long long ts2 = W::clock();
    ctr.cycles[18] += ts2 - ts1;
    {
      // Few live claims (the usual case once the dead row has filled in): the candidate the reference reaches first is
      // the live claim with the smallest position in its order (addToInflightNode, scheduler.go:667-686). Lane l owns
      // word l of the live set and takes the minimum of (position, claim) over its bits; one DPP reduction picks the
      // winner. No walk over the order at all.
      int who0;
      int densest;
      if constexpr (BIG) {
        // same test, over however many words the problem has: each lane counts the live words it owns, prefilter only when few
        int who1;
        const int my_words_max = 64 - (int)W::argmin_u32([&](int l) { int n = 0; for (int w = l; w < words; w += 64) n += stage[w] != 0; return (uint32_t)(64 - (n > 64 ? 64 : n)); }, &who1);
        if (my_words_max <= 1) {
          uint64_t nzl = W::ballot([&](int l) { for (int w = l; w < words; w += 64) if (stage[w]) return true; return false; });
          if (popc64(nzl) <= 8) {
            any = 0;
            for (; nzl; nzl &= nzl - 1) {
              const int l0 = ctz64(nzl);
              int w = -1;
              for (int x = l0; x < words; x += 64) if (stage[x]) { w = x; break; }   // uniform: stage is shared memory
              if (w < 0) continue;
              const uint64_t okm = W::ballot([&](int l) {
                const int c = w * 64 + l;
                if (c >= nc) return false;
                int ok = 1;
                for (int r = 0; r < nr; ++r) ok &= (int)(req[r] <= hd[(size_t)r * mc + c]);
                return ok != 0;
              });
              const uint64_t before = stage[w], v = before & okm;
              if (v != before) { W::store(&stage[w], v); W::store(&drow[w], (uint64_t)(drow[w] | (before & ~okm))); }
              any |= v;
            }
            W::sync();
            if (!any) return false;
          }
        }
        densest = 4096 - (int)W::argmin_u32([&](int l) { int n = 0; for (int w = l; w < words; w += 64) n += popc64(stage[w]); return (uint32_t)(4096 - (n > 4096 ? 4096 : n)); }, &who0);
      } else {
      {
        // Headroom prefilter, one lane per claim, 64 claims per ballot, over the words that still have live claims.
        uint64_t nz = W::ballot([&](int l) { return l < words && stage[l] != 0; });
        uint64_t nz_hi = words > 64 ? W::ballot([&](int l) { return l + 64 < words && stage[l + 64] != 0; }) : 0ull;
        if (popc64(nz) + popc64(nz_hi) <= 8) {   // more live words than that: the probes decide
          any = 0;
          while (nz | nz_hi) {
            int w;
            if (nz) { w = ctz64(nz); nz &= nz - 1; } else { w = 64 + ctz64(nz_hi); nz_hi &= nz_hi - 1; }
            const uint64_t okm = W::ballot([&](int l) {
              const int c = w * 64 + l;
              int64_t h[kRegNr];
#pragma unroll
              for (int r = 0; r < kRegNr; ++r) h[r] = r < nr ? hd[(size_t)r * mc + c] : INT64_MAX;
              int ok = 1;
#pragma unroll
              for (int r = 0; r < kRegNr; ++r) ok &= (int)(rq[r] <= h[r]);
              for (int r = kRegNr; r < nr; ++r) ok &= (int)(req[r] <= hd[(size_t)r * mc + c]);
              return ok != 0;
            });
            const uint64_t before = stage[w], v = before & okm;
            if (v != before) { W::store(&stage[w], v); W::store(&drow[w], (uint64_t)(drow[w] | (before & ~okm))); }
            any |= v;
          }
          W::sync();
          if (!any) return false;
        }
      }
      densest = 128 - (int)W::argmin_u32([&](int l) { return (uint32_t)(128 - (l < words ? popc64(stage[l]) : 0) - (l + 64 < words ? popc64(stage[l + 64]) : 0)); }, &who0);
      }
      if (densest <= (BIG ? 24 : 12)) {   // the per-lane loop below runs `densest` times (BIG: walking a long order in HBM costs more)
        for (;;) {
          int c;
          if constexpr (BIG) {
            const uint64_t best = W::reduce_min(64, [&](int l) -> uint64_t {
              uint64_t mine = ~0ull;
              for (int w = l; w < words; w += 64) for (uint64_t b = stage[w]; b; b &= b - 1) {
                const uint32_t cc = (uint32_t)(w * 64 + ctz64(b));
                const uint64_t key = ((uint64_t)order.position((int)cc) << 32) | cc;
                mine = key < mine ? key : mine;
              }
              return mine;
            });
            if (best == ~0ull) return false;
            c = (int)(uint32_t)best;
          } else {
            int who;
            const uint32_t best = W::argmin_u32([&](int l) {
              uint32_t mine = 0xFFFFFFFFu;
              for (int w = l; w < words; w += 64) for (uint64_t b = stage[w]; b; b &= b - 1) {
                const uint32_t cc = (uint32_t)(w * 64 + ctz64(b));
                const uint32_t key = (order.position((int)cc) << 13) | cc;
                mine = key < mine ? key : mine;
              }
              return mine;
            }, &who);
            if (best == 0xFFFFFFFFu) return false;
            c = (int)(best & 0x1FFFu);
          }
          ctr.cycles[19] += W::clock() - ts2;
          const int l = c >> 6;
          const uint64_t valid = (l == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
          const uint64_t live = stage[l] & ~(1ull << (c & 63));
          if (try_claim(k, c, pod) == E_OK) return true;
          unsigned long long tf = W::clock();
          if (!FULL || !cur_M) W::store(&drow[l], (uint64_t)(~live & valid));
          else if (!topo_reached) W::store(&drow[l], (uint64_t)(drow[l] | (1ull << (c & 63))));
          W::store(&stage[l], live);
          W::sync();
          ctr.cycles[8] += W::clock() - tf;
        }
      }
    }
    // Many live claims: walk the claims in the reference's order, 64 positions per ballot, testing the staged live bits:
    // the first live position is the first candidate; a failed probe clears its bit. The order is walked one SEGMENT at a
    // time — the whole array (LDS-resident order), or one ring per pod count, smallest count first (BIG: run_order.h; the
    // sort that precedes the scan leaves no pending move).
    auto walk = [&](int seg_n, auto claim_of) -> bool {
      for (int base0 = 0; base0 < seg_n; base0 += 512) {
        // 512 positions per step: eight independent (order -> live bit) gathers in flight, then eight ballots
        const int nchunks = (seg_n - base0 + 63) / 64 < 8 ? (seg_n - base0 + 63) / 64 : 8;
        uint64_t found = 0;   // first chunk with a live position
        int found_j = -1;
        W::ballots8(nchunks, [&](int l, int j) {
          int i = base0 + j * 64 + l;
          if (i >= seg_n) return false;
          uint32_t c = claim_of(i);
          return ((stage[c >> 6] >> (c & 63)) & 1) != 0;
        }, [&](int j, uint64_t m) { if (m && found_j < 0) { found_j = j; found = m; } });
        if (found_j < 0) continue;
        // probe from the first live position on, chunk by chunk (re-ballot after a failed probe: stage changed)
        for (int base = base0 + found_j * 64; base < seg_n && base < base0 + 512; base += 64) {
          uint64_t m = base == base0 + found_j * 64 ? found : W::ballot([&](int l) {
            int i = base + l;
            if (i >= seg_n) return false;
            uint32_t c = claim_of(i);
            return ((stage[c >> 6] >> (c & 63)) & 1) != 0;
          });
          while (m) {
            int i = base + ctz64(m);
            m &= m - 1;
            const int c = (int)W::uniform((uint64_t)claim_of(i));
            const int l = c >> 6;
            const uint64_t valid = (l == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
            const uint64_t live = stage[l] & ~(1ull << (c & 63));
            if (try_claim(k, c, pod) == E_OK) return true;
            unsigned long long tf = W::clock();
            // the class's dead word becomes: everything not live any more (closed claims may be recorded as dead too —
            // both are permanent until the column is reset), never touching bits of claims that do not exist yet. Classes
            // under topology constraints only record the failures that did not depend on domain counters.
            if (!FULL || !cur_M) W::store(&drow[l], (uint64_t)(~live & valid));
            else if (!topo_reached) W::store(&drow[l], (uint64_t)(drow[l] | (1ull << (c & 63))));
            W::store(&stage[l], live);
            W::sync();
            ctr.cycles[8] += W::clock() - tf;
          }
        }
      }
      return false;
    };
    if constexpr (BIG) {
      for (int kc = 1; kc <= order.max_cnt; ++kc) {
        const int sz = (int)W::uniform((uint64_t)order.size_(kc));
        if (!sz) continue;
        const uint32_t h = (uint32_t)W::uniform((uint64_t)order.head_(kc)), m = (uint32_t)W::uniform((uint64_t)order.mask_of(kc));
        const uint32_t* rg = order.ring + (uint32_t)W::uniform((uint64_t)order.off_(kc));
        if (walk(sz, [&](int i) { return rg[(h + (uint32_t)i) & m]; })) return true;
      }
      return false;
    } else {
      return walk(nc, [&](int i) { return order.claim_at(i); });
    }
  }
  // ---- new claim: addToNewNodeClaim (scheduler.go:695-790) --------------------------------------------------
  KS_DEV int add_to_new_claim(int k, int pod) {
    const int nr = lay.nr, iw = lay.iw, n_its = P.n_its;
    const RecLayout ly = lay;
    int first_err = 0, first_diag = 0;
    cadd(ctr.ref_bin_evaluations, (unsigned long long)((unsigned long long)n_claims));  // the reference tried every in-flight claim before coming here
    for (int t = 0; t < P.n_templates; ++t) {
      if (!((active_templates >> t) & 1)) continue;
      const uint64_t* trec = L.tmpl + (size_t)t * ly.c_hot_words();
      const uint64_t* tcold = L.tmpl_cold + (size_t)t * ly.cold_words();
      uint32_t lm = P.tmpl_limit_mask[t];
      const uint64_t* bin = trec;
      if (lm) {
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        if (((lm >> nr) & 1) && rem[nr] == 0) { if (!first_err) first_err = E_LIMITS; continue; }   // node limit — scheduler.go:711-715
        // filterByRemainingResources — scheduler.go:1069-1085 (instance types carry no "nodes" capacity, :1076)
        const ProblemView& Pv = P;
        const uint64_t* its = trec + ly.c_its();
        uint64_t any = 0;
        for (int w = 0; w < iw; ++w) {
          uint64_t in = its[w];
          uint64_t ok = in ? W::ballot([&](int l) {
            int it = w * 64 + l;
            if (it >= n_its || !((in >> l) & 1)) return false;
            bool v = true;
            for (int r = 0; r < nr; ++r) if ((lm >> r) & 1) v = v && Pv.it_cap[(size_t)r * n_its + it] <= rem[r];
            if ((lm >> nr) & 1) v = v && 0 <= rem[nr];
            return v;
          }) : 0;
          W::store(&sc.lim[w], ok);
          any |= ok;
        }
        W::sync();
        if (!any) { if (!first_err) first_err = E_LIMITS; continue; }
        // a copy of the template record with the limited instance types
        uint64_t* cl = sc.claim;
        const uint64_t* lim = sc.lim;
        W::for_n(ly.c_hot_words(), [&](int i) { cl[i] = (i >= ly.c_its() && i < ly.c_its() + iw) ? lim[i - ly.c_its()] : trec[i]; });
        bin = sc.claim;
      }
      host_seq++;  // NewNodeClaim draws a hostname-placeholder number for every attempt (nodeclaim.go:93)
      bool changed = false;
      cadd(ctr.ref_bin_evaluations, 1);
      int rc = can_add(bin, tcold, true, first_err == 0, &changed, nullptr, -1);
      if (rc == E_RESERVED) { last_diag = 0; return E_RESERVED; }   // voids the lower-weight templates (scheduler.go:736-751)
      if (rc != E_OK) { if (!first_err) { first_err = rc; first_diag = (rc == E_INSTANCE_TYPES || rc == E_MIN_VALUES) ? last_diag : 0; } continue; }
      if (n_claims >= S.max_claims || (!BIG && n_claims >= (S.probe ? S.pr_order_cap : P.lds.order_cap))) { W::store(S.status_out, 1); return -1; }
      int c = n_claims++;
      const uint32_t tm2 = hi32(trec[ly.c_meta2()]);
      uint64_t* o = sc.out;
      uint64_t* oc = sc.out_cold;
      if (!changed) {
        W::for_n(ly.rw, [&](int w) { o[w] = bin[w]; });
        if (W::leader()) { o[ly.c_f0()] = bin[ly.c_f0()]; o[ly.c_f1()] = bin[ly.c_f1()]; }
        W::for_n(ly.cold_words(), [&](int i) { oc[i] = tcold[i]; });
      }
      W::sync();
      const bool cold = sc.out[ly.c_f1()] != 0 || (tm2 & 2u);
      if (FULL && cur_rec) topo_record(sc.tmpl_taints[t & 31], out_ref(sc.out_cold), 0, c);
      if (FULL && P.reserved_on) commit_reservations(c, true);
      if (FULL && P.hp_on) W::store(&S.c_hp[c], cur_hp_use);
      uint32_t relaxed = 0;
      if (minv_lowered) {
        // karpenter.sh/nodeclaim-min-values-relaxed — scheduler.go:763-772
        const int32_t* nv = (const int32_t*)(sc.out_cold + 2 * ly.nk);
        const int32_t* ov_ = (const int32_t*)(tcold + 2 * ly.nk);
        for (int k = 0; k < ly.nk; ++k) if (ov_[k] >= 0 && nv[k] >= 0 && nv[k] < ov_[k]) relaxed = 1;
      }
      finish_record(c, bin, true, (uint32_t)t, 1u, host_seq, (tm2 & 2u) | relaxed, cold);

      order.append(c);
      if (lm) {
        // subtractMax — scheduler.go:1049-1066 : remaining -= max capacity over the claim's instance types
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        const uint64_t* sits = sc.its;
        const ProblemView& Pv = P;
        for (int r = 0; r < nr; ++r) if ((lm >> r) & 1) {
          int64_t mx = W::reduce_max_i64(n_its, [&](int it) { return ((sits[it >> 6] >> (it & 63)) & 1) ? Pv.it_cap[(size_t)r * n_its + it] : INT64_MIN; });
          W::store(&rem[r], rem[r] - mx);
        }
        W::sync();
      }
      commit_pod(pod, c, 0);
      return E_OK;
    }
    last_diag = first_diag;
    return first_err ? first_err : E_NO_TEMPLATES;
  }

  // ---- existing nodes: addToExistingNode (scheduler.go:614-656), ExistingNode.CanAdd/Add (existingnode.go:81-185) ----
  // 64 nodes per step, one lane per node, SoA loads coalesced across lanes; the lowest passing index wins
  // (scheduler.go:639). Strict Compatible (no AllowUndefinedWellKnownLabels, existingnode.go:100).
  //
  // A probe of a resident cluster (Workspace::probe) reads the cluster's pristine, shared node tables and keeps the nodes it
  // has committed pods to in its overlay (ov_*): the mutable n_* arrays are then indexed by overlay slot.
  KS_DEV int nst() const { return S.probe ? S.ov_cap : P.n_nodes; }   // stride of the mutable node arrays
  KS_DEV static uint32_t ov_hash(int e) { return (uint32_t)e * 2654435761u; }
  // slot of node e's mutable state: e itself outside probes, its overlay slot or -1 (pristine) in a probe. Per lane.
  KS_DEV int ov_find(int e) const {
    if (!S.probe) return e;
    if (ov_regs) {
      int os = -1;
      for (int s = 0; s < n_ov; ++s) if (ovk_.bcast(s) == (uint32_t)e + 1u) os = s;
      return os;
    }
    const uint32_t m = (uint32_t)S.ov_cap - 1;
    for (uint32_t h = (ov_hash(e) >> 8) & m;; h = (h + 1) & m) {
      const uint32_t kx = S.ov_key[h];
      if (kx == (uint32_t)e + 1u) return (int)h;
      if (!kx) return -1;
    }
  }
  KS_DEV NodeTabs node_tabs(bool overlay) const {
    NodeTabs t;
    if (overlay) { t.mask = S.n_mask; t.defined = S.n_defined; t.complement = S.n_complement; t.hg = S.n_hg; t.hl = S.n_hl; t.gte = S.n_gte; t.lte = S.n_lte; t.remaining = S.n_remaining; t.hp = P.hp_on ? S.n_hp : nullptr; t.stride = (size_t)nst(); }
    else { t.mask = S.n_mask0; t.defined = S.n_defined0; t.complement = S.n_complement0; t.hg = nullptr; t.hl = nullptr; t.gte = nullptr; t.lte = nullptr; t.remaining = S.n_remaining0; t.hp = P.node_hp0; t.stride = (size_t)P.n_nodes; }
    return t;
  }
  // a free overlay slot for node en (not overlaid yet); the caller fills the slot's state
  KS_DEV int ov_new(int en) {
    if (ov_regs) { const int os = n_ov; ovk_.set(os, (uint32_t)en + 1u); n_ov++; return os; }
    const uint32_t m = (uint32_t)S.ov_cap - 1;
    uint32_t h = (ov_hash(en) >> 8) & m;
    while (S.ov_key[h]) h = (h + 1) & m;
    if (W::leader()) S.ov_key[h] = (uint32_t)en + 1u;
    return (int)h;
  }
  // the overlay slot of node en, created from the pristine tables when the probe touches the node for the first time
  KS_DEV int ov_touch(int en) {
    int os = ov_find(en);
    if (os >= 0) return os;
    os = ov_new(en);
    const int ne = P.n_nodes, oc = S.ov_cap;
    const Workspace& Sw = S;
    const ProblemView& Pv = P;
    uint64_t* nm = S.n_mask; int64_t* nrem = S.n_remaining;
    W::for_n(lay.rw, [&](int w) { nm[(size_t)w * oc + os] = Sw.n_mask0[(size_t)w * ne + en]; });
    W::for_n(lay.nr, [&](int r) { nrem[(size_t)r * oc + os] = Sw.n_remaining0[(size_t)r * ne + en]; });
    if (W::leader()) {
      S.n_defined[os] = S.n_defined0[en]; S.n_complement[os] = S.n_complement0[en]; S.n_npods[os] = 0;
      if (S.n_hg) { S.n_hg[os] = 0; S.n_hl[os] = 0; }
      if (Pv.hp_on) S.n_hp[os] = Pv.node_hp0 ? Pv.node_hp0[en] : 0ull;
    }
    W::sync();
    return os;
  }
  // How many existing nodes before index `upto` the reference would have evaluated for this pod (probe mode): every node
  // that is part of the simulation and not skipped by the consolidateAfter rule (scheduler.go:628).
  KS_DEV unsigned long long probe_nodes_before(int upto, bool exempt_pod) const {
    long long n = upto;
    const int w = upto >> 6, bq = upto & 63;
    if (!exempt_pod && P.node_skip) n -= (long long)P.node_skip_prefix[w] + (bq ? popc64(P.node_skip[w] & ((1ull << bq) - 1)) : 0);
    if (prm_regs) {
      const uint64_t* sk = (!exempt_pod && P.node_skip) ? P.node_skip : nullptr;
      const uint32_t up = (uint32_t)upto;
      const LaneVar<uint32_t>& a0 = prm0_; const LaneVar<uint32_t>& a1 = prm1_;
      const uint64_t c0 = W::ballot([&](int l) { const uint32_t r = a0.v_of(l); return r < up && (!sk || !((sk[r >> 6] >> (r & 63)) & 1)); });
      const uint64_t c1 = W::ballot([&](int l) { const uint32_t r = a1.v_of(l); return r < up && (!sk || !((sk[r >> 6] >> (r & 63)) & 1)); });
      return (unsigned long long)(n - popc64(c0) - popc64(c1));
    }
    for (int i = 0; i < S.pr_n_removed; ++i) {
      const int r = (int)S.pr_removed[i];
      if (r >= upto) break;
      if (exempt_pod || !P.node_skip || !((P.node_skip[r >> 6] >> (r & 63)) & 1)) n--;
    }
    return (unsigned long long)n;
  }
  // probe mode: the next 64-node block after `base` that holds a node worth testing — not rejected in the pristine cluster
  // (n_dead0; an overlaid node only narrows, so the verdict stands unless a commit revived it), part of the simulation, not
  // skipped. The class's row is read 64 words (4096 nodes) per step. Returns the block's first node or -1; `todo` = its nodes.
  KS_DEV int probe_next_block(int k, int base, bool exempt_pod, int n_revived, uint64_t* todo) {
    const int ne = P.n_nodes, nw = P.node_words;
    const uint64_t* drow = P.n_dead0 + (size_t)k * nw;
    const uint64_t* skip = exempt_pod ? nullptr : P.node_skip;
    const Workspace& Sw = S;
    for (int w0 = (base >> 6) + 1; w0 < nw; w0 += 64) {
      cadd(ctr.node_block_steps, 1);
      LaneVar<uint64_t> lv, rmv;
      const bool rregs = prm_regs;
      if (rregs) {
        // the removed nodes that fall into this step's 64 words, each into its word's lane (usually none or a few)
        W::each([&](int l) { rmv.at(l) = 0; });
        const uint32_t wlo = (uint32_t)w0;
        for (int half = 0; half < 2; ++half) {
          const LaneVar<uint32_t>& a = half ? prm1_ : prm0_;
          for (uint64_t in = W::ballot([&](int l) { return ((a.v_of(l) >> 6) - wlo) < 64u; }); in; in &= in - 1) {
            const uint32_t r = a.bcast(ctz64(in));
            const int t = (int)((r >> 6) - wlo);
            W::each([&](int l) { if (l == t) rmv.at(l) |= 1ull << (r & 63); });
          }
        }
      }
      const uint64_t any = W::ballot([&](int l) {
        const int w = w0 + l;
        uint64_t live = 0;
        if (w < nw) {
          live = ~drow[w];
          for (int i = 0; i < n_revived; ++i) { const uint32_t r = Sw.pr_revived[i]; if ((int)(r >> 6) == w) live |= 1ull << (r & 63); }
          if (w == nw - 1 && (ne & 63)) live &= (1ull << (ne & 63)) - 1;
          if (skip) live &= ~skip[w];
          if (rregs) live &= ~rmv.at(l);
          else for (int i = 0; i < Sw.pr_n_removed; ++i) { const uint32_t r = Sw.pr_removed[i]; if ((int)(r >> 6) == w) live &= ~(1ull << (r & 63)); }
        }
        lv.at(l) = live;
        return live != 0;
      });
      if (any) { const int l = ctz64(any); *todo = lv.bcast(l); return (w0 + l) * 64; }
    }
    return -1;
  }
  // VolumeUsage.ExceedsLimits (volumeusage.go:193-200) for the pod being placed on existing node en: per driver with a limit
  // there, |volumes in use (before the solve + added by this solve's commits) ∪ the pod's| <= limit. One lane per volume of
  // the pod; `fresh` (out) = the pod's volumes that are not on the node yet, as a mask over its first 64 (more: never fresh
  // past the first block — pods mount a handful of volumes; a pod with more than 64 tracked volumes is refused upstream).
  KS_DEV bool node_exceeds_volume_limits(int en, uint64_t* fresh) {
    const ProblemView& Pv = P;
    const Workspace& Sw = S;
    const int nd = Pv.n_pv_drivers, nlog = n_pv_log;
    const uint32_t n0 = Pv.node_pv_first[en], n1 = Pv.node_pv_first[en + 1], pf = cur_pv_first, pn = cur_pv_n;
    const uint64_t want = W::ballot([&](int l) {
      if ((uint32_t)l >= pn) return false;
      const uint32_t v = Pv.pod_pvs[pf + l];
      uint32_t lo = n0, hi = n1;                      // binary search in the node's ascending list
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (Pv.node_pvs[mid] < v) lo = mid + 1; else hi = mid; }
      if (lo < n1 && Pv.node_pvs[lo] == v) return false;
      const uint64_t key = ((uint64_t)(uint32_t)en << 32) | v;
      for (int i = 0; i < nlog; ++i) if (Sw.pv_log[i] == key) return false;
      return true;
    });
    *fresh = want;
    bool exceeds = false;
    for (int dq = 0; dq < nd; ++dq) {
      const int32_t lim = Pv.node_pv_limit[(size_t)en * nd + dq];
      if (lim < 0) continue;
      int used = 0;
      for (uint32_t b = n0; b < n1; b += 64) used += popc64(W::ballot([&](int l) { return b + l < n1 && Pv.pv_driver[Pv.node_pvs[b + l]] == dq; }));
      for (int b = 0; b < nlog; b += 64) used += popc64(W::ballot([&](int l) { return b + l < nlog && (int)(Sw.pv_log[b + l] >> 32) == en && Pv.pv_driver[(uint32_t)Sw.pv_log[b + l]] == dq; }));
      used += popc64(W::ballot([&](int l) { return ((want >> l) & 1) && Pv.pv_driver[Pv.pod_pvs[pf + l]] == dq; }));
      exceeds = exceeds || used > lim;
    }
    return exceeds;
  }
  KS_DEV bool add_to_existing(int k, int pod) {
    const int ne = P.n_nodes;
    if (ne == 0) return false;
    const Dict& d = P.dict;
    const RecLayout ly = lay;
    const int nr = ly.nr;
    const Workspace& Sw = S;
    const ProblemView& Pv = P;
    const bool probe = S.probe != 0;
    const bool has_nz = ly.rw <= 64;
    uint64_t nz = 0;
    if (has_nz) { const uint64_t* cm_ = sc.cls + ly.k_mask(); const int rw_ = ly.rw; nz = W::ballot([&](int l) { return l < rw_ && cm_[l] != 0; }); }
    const NodeClassCtx cx = node_class_ctx(d, ly, sc.cls, sc.cls_cold, Pv.hp_on ? cur_hp_conf : 0ull, has_nz, nz);
    const int64_t* req = cx.req;
    const NodeTabs mut = node_tabs(true);
    const bool exempt_pod = cur_exempt;   // (fetched with the pod's queue block: pod_is_pending / pod_from_deleting)
    if (probe) W::sync();                 // behind the stores of the previous pod's commit (its first-pod path does not wait for them)
    uint64_t* ndead = probe ? nullptr : S.n_dead + (size_t)k * P.node_words;
    int base = -64;
    for (;;) {
      uint64_t todo, deadw = 0, validm = 0, skipped = 0;
      if (!probe) {
        base += 64;
        if (base >= ne) break;
        deadw = ndead[base >> 6];
        const int cnt = ne - base < 64 ? ne - base : 64;
        validm = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
        // nodes under consolidateAfter are skipped for non-pending pods that do not come from a deleting node (:628)
        const int b0 = base;
        skipped = exempt_pod ? 0ull : W::ballot([&](int l) { return l < cnt && (Pv.node_flags[b0 + l] & 2) != 0; });
        todo = validm & ~deadw & ~skipped;
        if (!todo) { cadd(ctr.ref_bin_evaluations, (unsigned long long)(popc64(validm & ~skipped))); continue; }
      } else {
        const unsigned long long tn0 = W::clock();
        base = probe_next_block(k, base, exempt_pod, n_revived, &todo);
        ctr.cycles[20] += W::clock() - tn0;
        if (base < 0) break;
      }
      const int b0 = base;
      const unsigned long long tn1 = W::clock();
      uint64_t ok, tov_ = 0;
      const bool lazy = probe;
      LaneVar<int> osv;                        // hash-table overlay: the slot of each lane's node (-1: pristine), asked once per block
      const bool osv_ok = probe && !ov_regs;
      if (lazy) {
        // A pristine node of the block that ksolve_node_dead0 left alive for this class PASSES these very checks (nodecheck.h is the
        // code of both, the class's host-port closure included), and a node a commit revived is an overlaid node: only the block's
        // overlaid nodes are evaluated (through the overlay) — the others' tables are not read at all until one of them is merged.
        uint64_t ovm = 0;
        if (ov_regs) {
          for (int s_ = 0; s_ < n_ov; ++s_) { const uint32_t e1 = ovk_.bcast(s_) - 1u - (uint32_t)b0; if (e1 < 64u) ovm |= 1ull << e1; }
        } else {
          ovm = W::ballot([&](int l) { const int os = ((todo >> l) & 1) ? ov_find(b0 + l) : -1; osv.at(l) = os; return os >= 0; });
        }
        ok = todo & ~ovm;
        const uint64_t tov = todo & ovm;
        tov_ = tov;
        const bool regs = ov_regs;
        if (tov) ok |= W::ballot([&](int l) {
          if (!((tov >> l) & 1)) return false;
          const int e_ = b0 + l;
          const size_t os = (size_t)(regs ? ov_find(e_) : osv.at(l));
          const NodePre n = node_preload(ly, Pv.node_taints[e_], mut, os);   // (the node's scalars in one round trip, not one per check)
          return node_static_ok_pre(d, ly, cx, n, mut, os);
        });
      } else {
        cadd(ctr.node_evaluations, (unsigned long long)popc64(todo));
        ok = W::ballot([&](int l) {
          if (!((todo >> l) & 1)) return false;
          const int e_ = b0 + l;
          return node_static_ok(d, ly, cx, Pv.node_taints[e_], mut, (size_t)e_);   // (outside probes the solve's own copies)
        });
      }
      const unsigned long long tn2 = W::clock();
      ctr.cycles[21] += tn2 - tn1;
      int l = -1;
      bool changed = false;
      LaneVar<uint64_t> npre; bool npre_ok = false;
      unsigned long long pnb = 0; bool pnb_ok = false;
      int os_hint = -2;                        // the winner's overlay slot where the block's lanes have looked it up already (-2: not known)
      ReqBuf* fin = &sc.merged;
      uint64_t pv_fresh = 0;
      if (!cur_M && !cur_vol_n && !cur_pv_n) {
        if (ok) l = ctz64(ok);
        if (l >= 0) {
          if (probe) { pnb = probe_nodes_before(base + l, exempt_pod); pnb_ok = true; }   // (its loads go out in front of node_merge's: one wait for both)
          if (osv_ok) os_hint = osv.bcast(l);
          changed = node_merge(base + l, probe ? &npre : nullptr, &npre_ok, os_hint);
        }
      } else {
        // volume requirement alternatives and topology decide among the nodes that passed everything else
        // (existingnode.go:108-139, tryVolumeAlternative :143-168), lowest index first
        for (uint64_t cand = ok; cand; cand &= cand - 1) {
          const int cl_ = ctz64(cand);
          if (cur_pv_n && node_exceeds_volume_limits(base + cl_, &pv_fresh)) continue;   // VolumeUsage.ExceedsLimits — existingnode.go:88
          if (osv_ok) os_hint = osv.bcast(cl_);
          const bool ch = node_merge(base + cl_, probe ? &npre : nullptr, &npre_ok, os_hint);
          bool tch = false, vch = false, got = false;
          cadd(ctr.bin_evaluations, 1);
          if (cur_vol_n) {
            reqbuf_copy(sc.vbase, sc.merged);
            for (uint32_t va = 0; va < cur_vol_n && !got; ++va) {
              if (va) reqbuf_copy(sc.merged, sc.vbase);
              const ReqRef vr = P.vol_reqs.at(d, cur_vol_first + va);
              if (reqs_compatible(d, sc.merged.ref(), vr, false) != COMPAT_OK) continue;           // existingnode.go:149-153
              vch = reqbuf_add(d, sc.merged, vr);
              W::sync();
              tch = false;
              if (cur_M && !topo_apply(sc.merged.ref(), 1, base + cl_, false, &tch)) continue;
              got = true;
            }
          } else got = !cur_M || topo_apply(sc.merged.ref(), 1, base + cl_, false, &tch);
          if (!got) continue;
          l = cl_; changed = ch || vch || tch;
          if (tch) fin = &sc.topo;
          break;
        }
      }
      const unsigned long long tn3 = W::clock();
      ctr.cycles[22] += tn3 - tn2;
      const uint64_t below = l < 0 ? ~0ull : (l == 0 ? 0ull : ((1ull << l) - 1));
      // nodes that failed a check that does not involve topology stay failed until their requirements change
      if (!probe) W::store(&ndead[base >> 6], (uint64_t)(deadw | (todo & ~ok & below)));
      if (l < 0) { if (!probe) cadd(ctr.ref_bin_evaluations, (unsigned long long)(popc64(validm & ~skipped))); else if (lazy) { cadd(ctr.node_evaluations, (unsigned long long)popc64(tov_)); cadd(ctr.bin_evaluations, (unsigned long long)popc64(tov_)); } continue; }
      const int en = base + l;
      if (!probe) cadd(ctr.ref_bin_evaluations, (unsigned long long)(popc64(validm & ~skipped & (below | (1ull << l)))));
      else cadd(ctr.ref_bin_evaluations, (unsigned long long)((pnb_ok ? pnb : probe_nodes_before(en, exempt_pod)) + 1));
      if (lazy) {   // (what was read: the overlaid nodes up to the winner, and the winner — its tables are read by node_merge)
        cadd(ctr.node_evaluations, (unsigned long long)(popc64(tov_ & below) + 1));
        cadd(ctr.bin_evaluations, (unsigned long long)(popc64(tov_ & below) + 1));
      } else cadd(ctr.bin_evaluations, (unsigned long long)(popc64(todo & (below | (1ull << l)))));
      // ---- ExistingNode.Add (existingnode.go:172-185): requirements <- node ∧ pod ∧ topology, remaining -= requests
      const ReqBuf& m = *fin;
      // A probe's first pod on a node whose requirements the pod leaves as they are (the usual case: the node's labels already say what
      // the pod asks for): the overlay slot is written straight from what is at hand — the merged set IS the node's, the remaining
      // resources are the pristine ones less the requests — instead of a copy of the pristine state that is then read back and updated
      // (two dependent round trips and a fence of the probe's chain).
      const bool fresh = probe && !changed && !S.n_hg && (os_hint != -2 ? os_hint : ov_find(en)) < 0;
      if (fresh) {
        const int os = ov_new(en);
        const size_t st = (size_t)nst();
        uint64_t* nm = S.n_mask; int64_t* nrem = S.n_remaining;
        const int64_t* rem0 = S.n_remaining0;
        W::for_n(ly.rw, [&](int w) { nm[(size_t)w * st + os] = m.mask[w]; });
        if (npre_ok) { const int r0_ = ly.rw + 2; W::each([&](int l) { if (l >= r0_ && l < r0_ + nr) nrem[(size_t)(l - r0_) * st + os] = (int64_t)npre.at(l) - req[l - r0_]; }); }   // (fetched with the node's requirement set)
        else W::for_n(nr, [&](int r) { nrem[(size_t)r * st + os] = rem0[(size_t)r * ne + en] - req[r]; });      // resources.SubtractFrom — existingnode.go:175
        if (W::leader()) {
          S.n_defined[os] = m.defined; S.n_complement[os] = m.complement; S.n_npods[os] = 1;
          if (Pv.hp_on) S.n_hp[os] = (Pv.node_hp0 ? Pv.node_hp0[en] : 0ull) | cur_hp_use;                  // existingnode.go:178
        }
        if (cur_rec) { W::sync(); topo_record(Pv.node_taints[en], m.ref(), 1, en); }                        // existingnode.go:184
        if (cur_pv_n && pv_fresh) {   // VolumeUsage.Add — existingnode.go:179
          const uint64_t fr = pv_fresh;
          const int at = n_pv_log;
          const uint32_t pf = cur_pv_first;
          uint64_t* lg = S.pv_log;
          W::each([&](int l) { if ((fr >> l) & 1) lg[at + popc64(fr & ((1ull << l) - 1))] = ((uint64_t)(uint32_t)en << 32) | Pv.pod_pvs[pf + l]; });
          n_pv_log += popc64(fr);
        }
        W::store(&S.assign[cur_out], (int32_t)(-2 - en));
        W::store(&S.slot[cur_out], 0u);
        W::order();   // (the fence stands in front of the next scan's reads of the overlay — the top of this function — where the stores' way to L2
                      // overlaps with the next pod's class fetch instead of being waited for here)
        ctr.cycles[23] += W::clock() - tn3;
        return true;
      }
      const int os = probe ? (os_hint >= 0 ? os_hint : ov_touch(en)) : en;     // the node's mutable state (a probe's overlay slot)
      const size_t st = (size_t)nst();
      uint64_t* nm = S.n_mask;
      if (changed) {
        // the same predicate as try_claim (node labels are single-valued In sets today, so only the defined-key half can fire)
        const uint32_t ex_b = exists_keys(Sw.n_complement[os], [&](int l) { return nm[(size_t)l * st + os]; });
        const uint32_t ex_a = ex_b ? exists_keys(m.complement, [&](int l) { return m.mask[l]; }) : 0;
        const bool revive = revives_rejections(Sw.n_defined[os], m.defined, ex_b, ex_a);
        W::for_n(ly.rw, [&](int w) { nm[(size_t)w * st + os] = m.mask[w]; });
        W::store(&S.n_defined[os], m.defined);
        W::store(&S.n_complement[os], m.complement);
        if (S.n_hg) {
          int64_t* ng = S.n_gte; int64_t* nl = S.n_lte;
          W::for_n(ly.nk, [&](int kk) { ng[(size_t)kk * st + os] = m.gte[kk]; nl[(size_t)kk * st + os] = m.lte[kk]; });
          W::store(&S.n_hg[os], m.has_gte);
          W::store(&S.n_hl[os], m.has_lte);
        }
        if (revive && !probe) {
          uint64_t* nd = S.n_dead;
          const int nwd = P.node_words;
          const uint64_t clr = ~(1ull << (en & 63));
          W::for_n(P.n_classes, [&](int kk) { nd[(size_t)kk * nwd + (en >> 6)] &= clr; });
        } else if (revive) {
          bool listed = false;
          for (int i = 0; i < n_revived; ++i) listed = listed || S.pr_revived[i] == (uint32_t)en;
          if (!listed) { W::store(&S.pr_revived[n_revived], (uint32_t)en); n_revived++; }
        }
      }
      if (cur_rec) topo_record(Pv.node_taints[en], m.ref(), 1, en);              // existingnode.go:184
      int64_t* nrem = S.n_remaining;
      W::for_n(nr, [&](int r) { nrem[(size_t)r * st + os] -= req[r]; });                  // resources.SubtractFrom — existingnode.go:175
      if (Pv.hp_on && cur_hp_use) W::store(&S.n_hp[os], (uint64_t)(S.n_hp[os] | cur_hp_use));   // existingnode.go:178
      if (cur_pv_n && pv_fresh) {   // VolumeUsage.Add — existingnode.go:179: the pod's volumes that the node did not hold yet
        const uint64_t fr = pv_fresh;
        const int at = n_pv_log;
        const uint32_t pf = cur_pv_first;
        uint64_t* lg = S.pv_log;
        W::each([&](int l) { if ((fr >> l) & 1) lg[at + popc64(fr & ((1ull << l) - 1))] = ((uint64_t)(uint32_t)en << 32) | Pv.pod_pvs[pf + l]; });
        n_pv_log += popc64(fr);
        W::sync();
      }
      const uint32_t np_ = S.n_npods[os];
      W::store(&S.n_npods[os], np_ + 1);
      W::store(&S.assign[cur_out], (int32_t)(-2 - en));
      W::store(&S.slot[cur_out], np_);
      W::sync();
      ctr.cycles[23] += W::clock() - tn3;
      return true;
    }
    if (probe) cadd(ctr.ref_bin_evaluations, (unsigned long long)(probe_nodes_before(ne, exempt_pod)));   // the reference evaluated every node of the simulation
    return false;
  }
  // sc.merged <- ExistingNode.requirements ∧ the pod's (existingnode.go:105-108); true when that differs from the node's
  // pre / pre_ok (probes): a PRISTINE node's remaining resources, fetched in the same round trip as its requirement set (lanes rw + 2 ..)
  // for the commit that usually follows (add_to_existing's first-pod path)
  KS_DEV bool node_merge(int en, LaneVar<uint64_t>* pre = nullptr, bool* pre_ok = nullptr, int os_hint = -2) {
    const Dict& d = P.dict;
    ReqBuf& m = sc.merged;
    const int os = os_hint != -2 ? os_hint : ov_find(en);
    const NodeTabs t = node_tabs(os >= 0);
    const size_t i = os >= 0 ? (size_t)os : (size_t)en;
    const int rw = lay.rw, nr = lay.nr;
    if (pre_ok) *pre_ok = false;
    if (pre && os < 0 && !t.hg && rw + 2 + nr <= 64) {
      // one round trip: the mask words, defined / complement flags and the remaining resources, a lane each
      LaneVar<uint64_t>& nv = *pre;
      W::each([&](int l) {
        uint64_t v = 0;
        if (l < rw) v = t.mask[(size_t)l * t.stride + i];
        else if (l == rw) v = t.defined[i];
        else if (l == rw + 1) v = t.complement[i];
        else if (l < rw + 2 + nr) v = (uint64_t)t.remaining[(size_t)(l - rw - 2) * t.stride + i];
        nv.at(l) = v;
      });
      W::each([&](int l) { if (l < rw) m.mask[l] = nv.at(l); });
      const uint32_t df = (uint32_t)nv.bcast(rw), cp = (uint32_t)nv.bcast(rw + 1);
      if (W::leader()) {
        m.defined = df; m.complement = cp; m.has_minv = 0; m.has_gte = 0u; m.has_lte = 0u;
        for (int kk = 0; kk < lay.nk; ++kk) { m.gte[kk] = 0; m.lte[kk] = 0; m.minv[kk] = -1; }
      }
      *pre_ok = true;
      // The usual outcome, decided wave-wide: the node's labels already say what the pod asks for — every key the pod defines is defined
      // on the node by a plain value set, no bounds or minValues anywhere, and no word of the node's masks narrows (In: a & b == a,
      // NotIn: a & ~b == a). Then Add (requirements.go:133-140) leaves the node's set as it is and reqbuf_add's key-by-key walk over
      // LDS — a chain of dependent reads per key — has nothing to do.
      const ReqRef q = class_ref(sc.cls, sc.cls_cold);
      if (!(q.has_gte | q.has_lte) && !q.minv && !(q.defined & ~df) && !(cp & q.defined)) {
        const uint8_t* wk = sc.word_key; const uint64_t* qm = q.mask;
        const uint32_t qd = q.defined, qc = q.complement;
        const uint64_t narrows = W::ballot([&](int l) {
          if (l >= rw) return false;
          const int k = wk[l];
          if (!((qd >> k) & 1)) return false;
          const uint64_t am = nv.at(l), b = qm[l];
          return (((qc >> k) & 1) ? (am & ~b) : (am & b)) != am;
        });
        if (!narrows) { W::sync(); return false; }
      }
    } else {
      W::for_n(rw, [&](int w) { m.mask[w] = t.mask[(size_t)w * t.stride + i]; });
      if (W::leader()) {
        m.defined = t.defined[i]; m.complement = t.complement[i]; m.has_minv = 0;
        m.has_gte = t.hg ? t.hg[i] : 0u; m.has_lte = t.hg ? t.hl[i] : 0u;
        for (int kk = 0; kk < lay.nk; ++kk) {
          m.gte[kk] = ((m.has_gte >> kk) & 1u) ? t.gte[(size_t)kk * t.stride + i] : 0;
          m.lte[kk] = ((m.has_lte >> kk) & 1u) ? t.lte[(size_t)kk * t.stride + i] : 0;
          m.minv[kk] = -1;
        }
      }
    }
    W::sync();
    ReqRef q = class_ref(sc.cls, sc.cls_cold);
    return reqbuf_add(d, sc.merged, q);
  }

  // ---- topology state of a probe of a resident cluster ------------------------------------------------------
  // The base problem's counts (TopoView::counts0 / node_counts0 / domains0) cover the whole cluster: every bound pod, every node.
  // A simulation counts the cluster WITHOUT its candidates and without the pods it is about to schedule (NewTopology puts them in
  // excludedPods, topology.go:92-94; countDomains :361-459 and updateInverseAffinities :310-355 skip them). So, after the copy:
  //   * a removed node: its per-node counters of hostname groups go, and it no longer registers its label values as domains;
  //   * a displaced pod (a row of this probe bound to a removed node): one less in every regular group that selects it on a
  //     node that passed the group's filter, and in every inverse anti-affinity group it owns;
  //   * the registered domains and the non-empty-domain counts of the dictionary-keyed groups are derived again.
  KS_DEV bool probe_node_removed(int e) const {
    if (prm_regs) {
      bool f = false;
      const int nrm = S.pr_n_removed;
      for (int i = 0; i < nrm; ++i) f = f || (i < 64 ? prm0_.bcast(i) : prm1_.bcast(i - 64)) == (uint32_t)e;
      return f;
    }
    for (int i = 0; i < S.pr_n_removed; ++i) if ((int)S.pr_removed[i] == e) return true;
    return false;
  }
  // sc.merged <- the pristine requirement set of existing node e (its labels); removed nodes are never overlaid
  KS_DEV void probe_load_node(int e) {
    ReqBuf& m = sc.merged;
    const int ne = P.n_nodes;
    const uint64_t* nm = S.n_mask0;
    W::for_n(lay.rw, [&](int w) { m.mask[w] = nm[(size_t)w * ne + e]; });
    if (W::leader()) {
      m.defined = S.n_defined0[e]; m.complement = S.n_complement0[e]; m.has_gte = m.has_lte = m.has_minv = 0;
      for (int kk = 0; kk < lay.nk; ++kk) { m.gte[kk] = 0; m.lte[kk] = 0; m.minv[kk] = -1; }
    }
    W::sync();
  }
  // the dictionary value a node's label gives key `key` (labels are single-valued In sets), -1 = the node has no such label
  KS_DEV int probe_node_value(int key) const {
    const Dict& d = P.dict;
    if (!bit(sc.merged.defined, key) || bit(sc.merged.complement, key)) return -1;
    for (uint32_t w = d.key_word_off[key]; w < d.key_word_off[key + 1]; ++w) if (sc.merged.mask[w]) return (int)(w - d.key_word_off[key]) * 64 + ctz64(sc.merged.mask[w]);
    return -1;
  }
  KS_DEV void probe_topology_adjust() {
    const TopoView& T = P.topo;
    const int G = T.n_groups, dv = T.dom_words * 64, ne = P.n_nodes;
    Workspace& Sw = S;
    if (T.dom_regs0) W::for_n(G * dv, [&](int i) { Sw.tg_regs[i] = T.dom_regs0[i]; });
    for (int ri = 0; ri < S.pr_n_removed; ++ri) {
      const int e = (int)S.pr_removed[ri];
      probe_load_node(e);
      const uint64_t taints = P.node_taints[e];
      for (int g = 0; g < G; ++g) {
        const int key = tt_key()[g];
        if (key < 0) {
          // the removed node's counter leaves with it (node_host_count reads 0 for it from here on)
          if (T.node_counts0[(size_t)tt_host_slot()[g] * ne + e] > 0) { W::store(&tg_N()[g], tg_N()[g] - 1); W::sync(); }
          continue;
        }
        if (!T.dom_regs0 || ((T.inverse_mask[g >> 6] >> (g & 63)) & 1)) continue;
        const int v = probe_node_value(key);
        if (v < 0 || !topo_filter_matches(g, taints, sc.merged.ref(), 1)) continue;
        int32_t* pr = S.tg_regs + (size_t)g * dv + v;
        W::store(pr, *pr - 1);
        W::sync();
      }
    }
    int loaded = -1;
    for (int i = 0; i < S.pr_n_pods; ++i) {
      const int pod = (int)S.pr_sorted[i];
      const int e = P.pod_node ? P.pod_node[pod] : -1;
      if (e < 0 || !probe_node_removed(e)) continue;
      if (loaded != e) { probe_load_node(e); loaded = e; }
      const uint64_t taints = P.node_taints[e];
      const uint64_t* ct = T.cls_topo + (size_t)P.row_class[pod] * 2 * T.words;
      for (int tw = 0; tw < T.words; ++tw)
      for (uint64_t todo = (ct[T.words + tw] & ~T.inverse_mask[tw]) | (ct[tw] & T.inverse_mask[tw]); todo; todo &= todo - 1) {
        const int g = tw * 64 + ctz64(todo);
        const int key = tt_key()[g];
        if (key < 0) continue;                                  // per-node counters of the removed node are gone already
        const bool inv = (T.inverse_mask[tw] >> (g & 63)) & 1;
        const int v = probe_node_value(key);
        if (v < 0) continue;
        if (!inv && !topo_filter_matches(g, taints, sc.merged.ref(), 1)) continue;
        int32_t* pc = tg_C() + (size_t)g * dv + v;
        W::store(pc, *pc - 1);
        W::sync();
      }
    }
    // NewTopology creates the groups of the pods it is given (topology.go:96-99): of the base problem's initially existing regular
    // groups only those a pod of THIS probe owns as submitted exist from the start; the others come into being if a pod relaxes
    // into them (fetch_class), and see no Record before that. Inverse groups come from the cluster's pods: all there.
    for (int tw = 0; tw < T.words; ++tw) {
      uint64_t own = 0;
      for (int i = 0; i < S.pr_n_pods; ++i) own |= T.cls_topo[(size_t)P.row_class[S.pr_sorted[i]] * 2 * T.words + tw];
      W::store(&sc.t_active[tw], (uint64_t)(T.initially_active[tw] & (T.inverse_mask[tw] | own)));
    }
    // registered domains (universe ∪ nodes that are still there ∪ counted) and the number of non-empty domains, per group
    for (int g = 0; g < G; ++g) {
      if (tt_key()[g] < 0) continue;
      const int32_t* cnt = tg_C() + (size_t)g * dv;
      const int32_t* rg = T.dom_regs0 ? S.tg_regs + (size_t)g * dv : nullptr;
      int nz = 0;
      for (int x = 0; x < T.dom_words; ++x) {
        const uint64_t counted = W::ballot([&](int b) { return cnt[x * 64 + b] > 0; });
        const uint64_t regd = rg ? W::ballot([&](int b) { return rg[x * 64 + b] > 0; }) : 0ull;
        nz += popc64(counted);
        if (T.dom_universe) W::store(&tg_D()[(size_t)g * T.dom_words + x], (uint64_t)(T.dom_universe[(size_t)g * T.dom_words + x] | regd | counted));
      }
      W::store(&tg_N()[g], nz);
    }
    W::sync();
  }

  // ---- class record of the pod being placed ----------------------------------------------------------------
  KS_DEV void fetch_class(int k) {
    const int hw = lay.k_hot_words();
    load_words(sc.cls, P.cls_hot + (size_t)k * hw, hw);
    if (FULL && (sc.cls[lay.k_f1()] != 0 || (lo32(sc.cls[lay.k_meta()]) & 1u))) load_words(sc.cls_cold, P.cls_cold + (size_t)k * lay.cold_words(), lay.cold_words());
    cur_class = k;
    if (FULL && P.hp_on) { cur_hp_use = P.cls_hp[(size_t)k * 2]; cur_hp_conf = P.cls_hp[(size_t)k * 2 + 1]; }
    if (FULL && P.vol_on) { const uint64_t v = P.cls_vol[k]; cur_vol_first = lo32(v); cur_vol_n = hi32(v); }

    if (FULL && P.topo.n_groups) {
      const TopoView& T = P.topo;
      const uint64_t* ct = T.cls_topo + (size_t)k * 2 * T.words;
      SC& s_ = sc;
      // getMatchingTopologies — topology.go:561-574: groups the pod owns + inverse anti-affinity groups that select it;
      // Topology.Update creates the groups of a relaxed pod (topology.go:162-194)
      const uint64_t anyM = W::ballot([&](int w) {
        if (w >= T.words) return false;
        const uint64_t ow = ct[w], se = ct[T.words + w], inv = T.inverse_mask[w];
        const uint64_t mt = (ow & ~inv) | (se & inv);
        s_.t_owned[w] = ow; s_.t_sel[w] = se; s_.t_match[w] = mt; s_.t_active[w] |= ow & ~inv & ~T.alias_mask[w];
        return mt != 0;
      });
      if (T.n_alias) {
        // Topology.Update looks a relaxed pod's group up by TopologyGroup.Hash() (topology.go:162-194): of the same-hash
        // groups different pods would build, the one created first exists and every later owner joins it
        W::sync();
        for (int w = 0; w < T.words; ++w) for (uint64_t m = ct[w] & T.alias_mask[w]; m; m &= m - 1) {
          const int g = w * 64 + ctz64(m);
          int32_t* slot = S.tg_alias_active + T.alias_class[g];
          const int a = *slot;
          const uint64_t gb = 1ull << (g & 63);
          if (a < 0) {
            W::store(slot, (int32_t)g);
            W::store(&s_.t_active[w], (uint64_t)(s_.t_active[w] | gb));
          } else if (a != g) {
            W::store(&s_.t_owned[w], (uint64_t)(s_.t_owned[w] & ~gb));
            W::store(&s_.t_match[w], (uint64_t)(s_.t_match[w] & ~gb));
            W::sync();
            const uint64_t ab = 1ull << (a & 63);
            W::store(&s_.t_owned[a >> 6], (uint64_t)(s_.t_owned[a >> 6] | ab));
            W::store(&s_.t_match[a >> 6], (uint64_t)(s_.t_match[a >> 6] | ab));
          }
          W::sync();
        }
      }
      const uint64_t anyR = W::ballot([&](int w) { return w < T.words && (ct[w] | ct[T.words + w]) != 0; });
      W::sync();
      cur_M = anyM != 0; cur_rec = anyR != 0;
    }
  }

  // add — scheduler.go:582-612
  KS_DEV int add_class(int k, int pod) {
    if (FULL && add_to_existing(k, pod)) return E_OK;  // scheduler.go:594
    cadd(ctr.sorts, 1);
    unsigned long long t0 = W::clock();
    order.sort();                                      // scheduler.go:598
    if constexpr (BIG) { if (order.overflow) { W::store(S.status_out, 1); return -1; } }   // a claim with more pods than the ring tables provide for: capacity
    unsigned long long t1 = W::clock();
    ctr.cycles[2] += t1 - t0;
    bool ok = scan_inflight(k, pod);                   // scheduler.go:601
    ctr.cycles[3] += W::clock() - t1;
    if (ok) return E_OK;
    if (active_templates == 0) { last_diag = 0; return E_NO_TEMPLATES; }   // scheduler.go:604-606
    t1 = W::clock();
    int rc = add_to_new_claim(k, pod);                 // scheduler.go:607
    ctr.cycles[7] += W::clock() - t1;
    return rc;
  }
  // trySchedule — scheduler.go:521-552 ; the relaxation ladder (preferences.go:38-57) is precomputed as a row chain
  KS_DEV int try_schedule(int pod, int k0) {
    int row = pod;
    int k = k0;
    for (;;) {
      unsigned long long t0 = W::clock();
      fetch_class(k);
      ctr.cycles[1] += W::clock() - t0;
      int rc = add_class(k, pod);
      if (rc == E_OK || rc < 0) return rc;
      if (rc == E_RESERVED) return rc;
      int nxt = P.row_next[row];
      if (nxt < 0) return rc;
      row = nxt;
      k = (int)P.row_class[row];
      cadd(ctr.relaxations, 1);
    }
  }

  // NewScheduler's per-template prefilter (scheduler.go:156-171): instance types compatible with the template's own
  // requirements, with non-negative allocatable and a compatible available offering. Templates become claim-shaped
  // records in LDS (total 0, unlimited headroom).
  // warm: the template records in LDS (L.tmpl / L.tmpl_cold) and `warm_active` come from another solve of the same problem (the
  // compact sweep: the prefilter does not depend on the probe); only this solve's own copies are written
  KS_DEV void prefilter_templates(bool warm = false, uint32_t warm_active = 0) {
    const Dict& d = P.dict;
    active_templates = 0;
    const int nr = lay.nr, iw = lay.iw;
    const RecLayout ly = lay;
    if (warm) {
      active_templates = warm_active;
      for (int t = 0; t < P.n_templates; ++t) {
        const uint64_t* rec = L.tmpl + (size_t)t * ly.c_hot_words();
        uint64_t* tits = S.t_its + (size_t)t * iw;
        W::for_n(iw, [&](int w) { tits[w] = rec[ly.c_its() + w]; });
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        const int64_t* lim = ((S.probe && S.pr_limits) ? S.pr_limits : P.tmpl_limits) + (size_t)t * (nr + 1);
        W::for_n(nr + 1, [&](int r) { rem[r] = lim[r]; });
      }
      return;
    }
    for (int r = 0; r < nr; ++r) W::store(&sc.total[r], (int64_t)0);
    W::sync();
    for (int t = 0; t < P.n_templates; ++t) {
      ReqRef tr = P.tmpl_reqs.at(d, t);
      uint64_t* rec = L.tmpl + (size_t)t * ly.c_hot_words();
      uint64_t* cold = L.tmpl_cold + (size_t)t * ly.cold_words();
      W::for_n(ly.rw, [&](int w) { rec[w] = tr.mask[w]; });
      int64_t* cg = (int64_t*)cold; int64_t* cl = cg + ly.nk; int32_t* cv = (int32_t*)(cold + 2 * ly.nk);
      bool has_minv = false;
      for (int k = 0; k < ly.nk; ++k) if (tr.minv && tr.minv[k] >= 0) has_minv = true;
      W::for_n(ly.nk, [&](int k) { cg[k] = (tr.gte && bit(tr.has_gte, k)) ? tr.gte[k] : 0; cl[k] = (tr.lte && bit(tr.has_lte, k)) ? tr.lte[k] : 0; cv[k] = tr.minv ? tr.minv[k] : -1; });
      if (W::leader()) {
        rec[ly.c_f0()] = (uint64_t)tr.defined | ((uint64_t)tr.complement << 32);
        rec[ly.c_f1()] = (uint64_t)tr.has_gte | ((uint64_t)tr.has_lte << 32);
        rec[ly.c_meta()] = (uint64_t)(uint32_t)t;
        rec[ly.c_meta2()] = (uint64_t)(has_minv ? 2u : 0u) << 32;
        for (int r = 0; r < nr; ++r) { rec[ly.c_total() + r] = 0; rec[ly.c_head() + r] = (uint64_t)INT64_MAX; }
      }
      W::sync();
      ReqRef rr = claim_ref(rec, cold);
      bool any = filter_instance_types(P.tmpl_its + (size_t)t * iw, sc.total, true, rr, false, -1);
      if (FULL && any && has_minv) {
        // scheduler.go:159: the prefilter applies minValues too; BestEffort keeps the template without touching its requirements
        int32_t* tmp = (int32_t*)(sc.out_cold + 2 * ly.nk);
        W::for_n(ly.nk, [&](int k) { tmp[k] = cv[k]; });
        bool lowered = false;
        if (!min_values_ok(tmp, min_values_best_effort, &lowered)) any = false;
      }
      const uint64_t* sits = sc.its;
      uint64_t* tits = S.t_its + (size_t)t * iw;
      W::for_n(iw, [&](int w) { rec[ly.c_its() + w] = sits[w]; tits[w] = sits[w]; });
      if (any) active_templates |= 1u << t;
      int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
      const int64_t* lim = ((S.probe && S.pr_limits) ? S.pr_limits : P.tmpl_limits) + (size_t)t * (nr + 1);
      W::for_n(nr + 1, [&](int r) { rem[r] = lim[r]; });
    }
  }

  // The compact sweep's once-per-workgroup half of solve(): the shared LDS tables and the template records (the prefilter does not
  // depend on the probe; the Workspace passed to this engine only lends its per-template arrays). Returns the surviving templates.
  KS_DEV uint32_t prepare() {
    load_tables(true);
    prefilter_templates();
    W::sync();
    return active_templates;
  }

  // Solve — scheduler.go:440-519 with Queue (queue.go:31-108)
  // warm != nullptr: the compact sweep — the LDS tables shared by the workgroup's wavefronts and the template records are in
  // place (*warm = the templates that survived the prefilter); this solve sets up its own working set only.
  KS_DEV void solve(const uint32_t* warm = nullptr) {
    const unsigned long long t_begin = W::clock();
    if (FULL && P.n_nodes && S.probe) {
      // a probe of a resident cluster: nothing is copied, the overlay starts empty
      Workspace& Sw = S;
      ov_regs = S.ov_cap == 64; n_ov = 0;
      prm_regs = S.pr_n_removed <= 128;
      if (prm_regs) { const uint32_t* pr = S.pr_removed; const int nrm = S.pr_n_removed; W::each([&](int l) { prm0_.at(l) = l < nrm ? pr[l] : 0xFFFFFFFFu; prm1_.at(l) = 64 + l < nrm ? pr[64 + l] : 0xFFFFFFFFu; }); }
      if (!ov_regs) W::for_n(S.ov_cap, [&](int i) { Sw.ov_key[i] = 0; });
      n_revived = 0;
    } else if (FULL && P.n_nodes) {
      // ExistingNodes are mutated by Solve: start from the pristine copies
      const int ne = P.n_nodes;
      Workspace& Sw = S;
      W::for_n(lay.rw * ne, [&](int i) { Sw.n_mask[i] = Sw.n_mask0[i]; });
      W::for_n(lay.nr * ne, [&](int i) { Sw.n_remaining[i] = Sw.n_remaining0[i]; });
      W::for_n(ne, [&](int i) { Sw.n_defined[i] = Sw.n_defined0[i]; Sw.n_complement[i] = Sw.n_complement0[i]; Sw.n_npods[i] = 0; });
      if (P.hp_on) { const uint64_t* h0 = P.node_hp0; W::for_n(ne, [&](int i) { Sw.n_hp[i] = h0 ? h0[i] : 0ull; }); }
      if (Sw.n_hg) W::for_n(ne, [&](int i) { Sw.n_hg[i] = 0; Sw.n_hl[i] = 0; });
    }
    topo_to_lds();
    if (FULL && P.topo.n_groups) {
      const TopoView& T = P.topo;
      Workspace& Sw = S;
      const int G = T.n_groups, dv = T.dom_words * 64;
      W::for_n(G * T.dom_words, [&](int i) { tg_D()[i] = T.domains0[i]; });
      W::for_n(G * dv, [&](int i) { tg_C()[i] = T.counts0[i]; });
      if (!S.probe) W::for_n(T.n_host_groups * P.n_nodes, [&](int i) { Sw.tg_node_counts[i] = T.node_counts0[i]; });   // probes: shared + overlay (node_host_count)
      W::for_n(G, [&](int i) { tg_N()[i] = T.nonzero0[i]; });
      W::for_n(T.words, [&](int w) { sc.t_active[w] = T.initially_active[w]; });
      W::for_n(T.n_alias, [&](int i) { Sw.tg_alias_active[i] = -1; });
      if (S.probe) probe_topology_adjust();
    }
    n_pv_log = 0;
    load_tables(warm == nullptr);
    if (FULL && P.reserved_on) W::for_n(P.n_resv, [&](int i) { sc.resv_cap[i] = P.resv_cap0[i]; });
    prefilter_templates(warm != nullptr, warm ? *warm : 0u);
    // The queue holds pod indices; a probe's queue holds positions in its own pod list (Workspace::pr_sorted), which is also
    // how its per-pod outputs are indexed.
    // (the loop's own state as wave-uniform scalars: read through the view / workspace references — LDS copies in the sweep kernels — these
    // values come back in vector registers and stayed there across every pod, first in line to be spilled; no division by a run-time
    // value either: the ring positions wrap with a compare)
    const bool probe = W::uniform((uint64_t)(S.probe != 0)) != 0;
    const int np = (int)W::uniform((uint64_t)(uint32_t)(probe ? S.pr_n_pods : P.n_pods));
    const uint32_t cap = (uint32_t)np + 1;
    const uint32_t* sorted = (const uint32_t*)W::uniform((uint64_t)(probe ? S.pr_sorted : P.sorted_pods));
    uint32_t* queue = (uint32_t*)W::uniform((uint64_t)S.queue);
    W::for_n(np, [&](int i) { queue[i] = probe ? (uint32_t)i : sorted[i]; });
    uint32_t head = 0, tail = (uint32_t)np == cap ? 0u : (uint32_t)np, qlen = (uint32_t)np;
    long long steps = 0;
    int status = 0;
    int blk_n = 0, blk_i = 0;
    while (qlen > 0) {
      unsigned long long tq = W::clock();
      if (blk_i >= blk_n) {
        // fetch the next (up to) 64 queue entries with their class ids and lastLen in two coalesced round trips
        blk_n = qlen < 64 ? (int)qlen : 64;
        blk_i = 0;
        uint32_t* bp = sc.blk_pod; uint32_t* bc = sc.blk_class; uint32_t* bl = sc.blk_last; uint32_t* bo = sc.blk_out;
        const uint32_t* rc_ = P.row_class; const uint32_t* ll = S.last_len;
        const uint8_t* pend = (FULL && P.n_nodes) ? P.pod_is_pending : nullptr; const uint8_t* pdel = (FULL && P.n_nodes) ? P.pod_from_deleting : nullptr;
        const int bn = blk_n;
        const uint32_t h0 = head;
        W::for_n(64, [&](int l) {
          if (l < bn) {
            uint32_t qi = h0 + (uint32_t)l;   // (< 2 cap: head < cap and l < the entries there are < cap)
            if (qi >= cap) qi -= cap;
            const uint32_t q = queue[qi]; const uint32_t p = probe ? sorted[q] : q; bp[l] = p; bo[l] = q; bc[l] = rc_[p]; bl[l] = ll[q] | (((pend && pend[p]) || (pdel && pdel[p])) ? 0x80000000u : 0u);   // (bit 31: exempt from the consolidateAfter skip)
          }
        });
        if (S.cancel_flag) {   // ctx cancellation, polled once per 64 pods (< 0: tests only, see fast_engine.h)
          const int cv = W::poll_flag(S.cancel_flag);
          if (cv > 0 || (cv < 0 && (long long)steps >= -(long long)cv)) { status = 2; break; }
        }
      }
      // (wave-uniform scalars, not the vector registers an LDS read comes back in: they live across the whole placement of the pod)
      int pod = (int)W::uniform((uint64_t)sc.blk_pod[blk_i]);
      const int out = (int)W::uniform((uint64_t)sc.blk_out[blk_i]);
      const uint32_t lastw = (uint32_t)W::uniform((uint64_t)sc.blk_last[blk_i]);
      cur_out = out;
      if (FULL && P.pv_on) { cur_pv_first = (uint32_t)W::uniform((uint64_t)P.pod_pv_first[pod]); cur_pv_n = (uint32_t)W::uniform((uint64_t)P.pod_pv_first[pod + 1]) - cur_pv_first; }
      cur_exempt = (lastw >> 31) != 0;
      if ((lastw & 0x7FFFFFFFu) == qlen) break;                // queue.go:52-56
      if (S.max_steps >= 0 && steps >= S.max_steps) { status = 2; break; }
      int k0 = (int)W::uniform((uint64_t)sc.blk_class[blk_i]);
      blk_i++;
      head = head + 1 == cap ? 0u : head + 1; qlen--;
      steps++;
      cadd(ctr.queue_pops, 1);
      ctr.cycles[0] += W::clock() - tq;
      unsigned long long ts = W::clock();
      int rc = try_schedule(pod, k0);
      ctr.cycles[9] += W::clock() - ts;
      if (rc < 0) { status = 1; break; }
      if (rc != E_OK) {
        W::store(&S.err[out], (uint8_t)rc);
        W::store(&S.diag[out], (uint8_t)last_diag);
        W::store(&S.queue[tail], (uint32_t)out);
        tail = tail + 1 == cap ? 0u : tail + 1; qlen++;
        W::store(&S.last_len[out], qlen);                                   // queue.go:63-66
        W::sync();
      } else {
        W::store(&S.err[out], (uint8_t)0);
        W::store(&S.diag[out], (uint8_t)0);
      }
    }
    ctr.slow_sorts = order.slow_sorts;
    ctr.cycles[10] = W::clock() - t_begin;
    {
      // results read the final order from HBM
      if constexpr (!BIG) {
        uint32_t* go = S.o_ord;
        const KS_LDS uint32_t* lo_ = L.oord;
        W::for_n(n_claims, [&](int i) { go[i] = lo_[i]; });
      } else order.write_final();
    }
    W::store(S.n_claims_out, n_claims);
    if (status) W::store(S.status_out, status);
    if (W::leader()) *S.counters = ctr;
    W::sync();
  }
};

}  // namespace ks
