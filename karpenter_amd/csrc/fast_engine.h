// fast_engine.h — the cursor engine: Scheduler.Solve() (scheduler.go:440-519) for provisioning batches whose
// requirement algebra is purely positive (every operator In — nodeSelector / node-affinity In terms, NodePool In
// requirements, instance types that label themselves with In sets), no topology, no existing nodes, no daemon overhead,
// no minValues, no reservations: BASELINE configs[0], [1] and [3]. One wavefront per scheduling problem, like engine.h,
// but built on three facts that hold for this shape and make a step O(1) instead of a scan over the claims:
//
//  1. A claim's InstanceTypeOptions are a pure function of its requirement set and its requests:
//         its = F(template, requirements) ∩ { it : allocatable(it) >= requests }          (nodeclaim.go:541-638)
//     because both only ever narrow / grow (NodeClaim.Add, nodeclaim.go:247-263) and with positive sets
//     Intersects (requirements.go:254-274) is monotone. So CanAdd (nodeclaim.go:124-242) needs no per-claim instance-type
//     mask: "some type of F still fits" is a dominance test against the Pareto-maximal allocatable vectors of F, which are
//     cached per distinct requirement set (a few hundred per problem) in LDS. The masks themselves are materialised once,
//     after the loop, by ksolve_fast_records (one wavefront per claim).
//  2. The requirement set of a claim is the template's plus a handful of keys pods select on; the values of those keys are
//     packed into ONE 64-bit word per claim (`vmask`: bit = value still allowed; after every key's field one guard bit,
//     set while the requirement set does not define the key), so Requirements.Compatible + Add (requirements.go:181-197,
//     133-140) is an AND, an ADD and a compare in registers.
//  3. CanAdd failures are permanent (fact 1), claims only move RIGHT in the reference's order when they gain a pod
//     (sort.Slice by pod count, scheduler.go:598 — pdq_emul.h keeps Go's exact permutation), and a new claim enters at
//     one known position. So each pod class keeps a cursor: "every claim left of it has rejected this class for good".
//     addToInflightNode's "lowest index that accepts" (scheduler.go:667-686) is then the first acceptor at or after the
//     cursor — in the steady state the claim AT the cursor — found by testing 64 positions per step, one lane each.
//     Cursors live in vector registers (lane = class slot) and are kept valid under moves with three VALU ops.
//
// The hot loop touches LDS and registers only (claim state, order, requirement-set cache, class slots); HBM sees the
// queue (64 pods per fetch) and the two result stores per pod.
//
// Everything this engine does not handle (an unschedulable pod, NodePool limits that actually exclude a type, more claims
// than the LDS plan holds, non-positive operators, ...) makes it stop with status 3 before it has written a result;
// the host then runs the general engine (engine.h) on the same problem. There is no CPU path.
#pragma once
#include <type_traits>
#include "kernels.h"
#include "pdq_emul.h"

namespace ks {

constexpr int kFastRows = 4;          // class slots = 64 lanes x kFastRows registers
constexpr int kFastSlots = 64 * kFastRows;
constexpr int kFastEnt = 1024;        // requirement-set cache slots (open addressing, filled to 80% at most)
constexpr int kFastPool = 256;        // extra Pareto vectors
constexpr int kFastMaxPareto = 16;    // per requirement set
constexpr int kFastMaxVar = 12;       // keys pods select on
constexpr int kFastClasses = 8192;    // pod classes per problem (class -> slot table in LDS)
constexpr int kFastVarBits = 56;      // their dictionary values + one guard bit each must fit 56 bits; the top byte of vmask is the template

struct FastClaim { uint64_t vmask; int32_t req[4]; };                                   // 24 B, LDS, by claim id
// a pod class as the scan needs it: values it admits (all ones on keys it does not select on), the fields it selects on,
// requests, templates whose taints it tolerates and whose keys cover its custom keys, keys it defines
struct FastSlot { uint64_t cvmask; uint64_t dmask; int32_t size[4]; uint32_t tmplok; uint32_t kdef; };  // 40 B
struct FastEnt { uint64_t vmask; int32_t cap[4]; uint32_t info; uint32_t pad; };        // 32 B: info bit0 valid, bits 8..15 extra vectors, bits 16..31 pool offset

struct FastPlan {   // LDS plan of ksolve_pack_fast (bytes), computed by the host
  int total_bytes, cap;
  int off_state, off_key, off_ord, off_snap, off_ent, off_pool, off_slot, off_misc, off_hot;
  int global_state;   // 1: the claims' state (FastClaim, 24 B each) lives in HBM (FastWork::c_state), only the order arrays in LDS:
                      //    ~15,000 in-flight claims instead of ~3,000 (round 4). off_state is unused then.
                      // 2: the order arrays too (FastWork::o_key / o_ord / o_snap): 65,472 claims — what 16-bit claim ids address;
                      //    LDS holds the caches, the class slots and the loop's own state only.
};

struct FastMisc {   // small LDS tables
  uint64_t tvmask[32];                 // template: values it admits on the variable keys | template << 56
  uint32_t tdef[32];
  uint64_t fmask[kFastMaxVar];         // field of variable key j inside vmask
  uint8_t vkey[kFastMaxVar], voff[kFastMaxVar], vwidth[kFastMaxVar];
  uint16_t vword[kFastMaxVar];         // dictionary word of the key
  uint64_t its[kMaxItWords], rem[kMaxItWords], cand[kMaxItWords];   // slow-path scratch
  uint32_t blk_pod[64];                // the 64 queue entries being placed
  uint16_t blk_class[64];
  uint32_t out_claim[64];              // their results, written to HBM when the block is done
  uint32_t out_cnt[64];
  uint16_t active[kFastSlots];         // class of each slot (for eviction)
  uint16_t slot_of[kFastClasses];      // class -> slot, 0xFFFF = none
  uint64_t acc[kFastRows];             // class slots that accept the claim just created (place_new_claim)
};

struct FastVar { int nv; uint8_t vkey[kFastMaxVar], voff[kFastMaxVar], vwidth[kFastMaxVar]; uint16_t vword[kFastMaxVar]; };   // the keys pods select on

struct FastWork {   // HBM workspace of the cursor engine (host-allocated when the problem may qualify)
  FastVar* var;           // written by the pack kernel, read by ksolve_fast_records
  FastSlot* cls;          // [n_classes]
  uint32_t* c_hostseq;    // [max_claims]
  uint16_t* c_ent;        // [max_claims] cache entry of the claim's requirement set
  FastClaim* c_state;     // [max_claims] final state, written when the loop ends
  uint32_t* c_npods;      // [max_claims]
  uint64_t* ent_its;      // [kFastEnt][it_words] F(requirement set)
  // The loop reads the queue and writes its results in QUEUE order, 64 consecutive entries per access: one wave touching
  // 64 random pods per block pays for 64 address translations in a row. ksolve_fast_queue (before) and
  // ksolve_fast_scatter (after) do the random accesses with every CU busy.
  uint32_t* q_class;      // [n_pods] class of queue entry i = row_class[sorted_pods[i]]
  uint32_t* q_claim;      // [n_pods] claim of queue entry i (0xFFFFFFFF: not placed)
  uint32_t* q_cnt;        // [n_pods] pods the claim held before it
  uint16_t* o_key;        // [max_claims] plan 2: the claim order (pod count / claim id by position) and its snapshot, in HBM
  uint16_t* o_ord;
  uint16_t* o_snap;
  FastPlan plan;
  int enabled;
};

// whole-record moves between LDS and registers (a struct behind an address-space-3 pointer has no implicit copy)
typedef uint64_t __attribute__((may_alias)) u64_alias;   // the records are moved as 8-byte words whatever their field types
template <class T>
KS_FN T lds_get(const KS_LDS T* p) {
  static_assert(sizeof(T) % 8 == 0, "lds_get: 8-byte multiples");
  T out;
  u64_alias* o = (u64_alias*)&out;
  const KS_LDS u64_alias* s = (const KS_LDS u64_alias*)p;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 8); ++i) o[i] = s[i];
  return out;
}
template <class T>
KS_FN void lds_put(KS_LDS T* p, const T& v) {
  static_assert(sizeof(T) % 8 == 0, "lds_put: 8-byte multiples");
  const u64_alias* o = (const u64_alias*)&v;
  KS_LDS u64_alias* s = (KS_LDS u64_alias*)p;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 8); ++i) s[i] = o[i];
}


// The in-flight claims' state by claim id. LDS while the problem's claims fit beside the order arrays and the caches (the
// benchmarked configuration: 2,763 claims); HBM otherwise (GS = true): the lane that tests a claim then gathers its 24 bytes
// through the vector L1 / L2 instead of LDS. Plain loads and stores: this wavefront is the only reader and writer, its vector
// memory operations execute in order, W::sync() (a wavefront-scope fence) follows every store — what the general engine's
// HBM-resident claim records have relied on since round 1. (-DKS_CLAIM_STATE_AGENT_SCOPE: relaxed atomics at agent scope
// instead, every access served by L2 — the first build of the plan, 1.48 µs per pod at 2M pods of configs[1].)
template <bool GS> struct ClaimStates;
template <> struct ClaimStates<false> {
  KS_LDS FastClaim* p;
  KS_FN FastClaim get(uint32_t c) const { return lds_get(&p[c]); }
  KS_FN void put(uint32_t c, const FastClaim& v) const { lds_put(&p[c], v); }
};
template <> struct ClaimStates<true> {
  FastClaim* p;
  KS_FN FastClaim get(uint32_t c) const {
    FastClaim out;
    u64_alias* o = (u64_alias*)&out;
#if KS_DEVICE && defined(KS_CLAIM_STATE_AGENT_SCOPE)
    const uint64_t* s = (const uint64_t*)&p[c];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FastClaim) / 8); ++i) o[i] = __hip_atomic_load(&s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    const u64_alias* s = (const u64_alias*)&p[c];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FastClaim) / 8); ++i) o[i] = s[i];
#endif
    return out;
  }
  KS_FN void put(uint32_t c, const FastClaim& v) const {
    const u64_alias* o = (const u64_alias*)&v;
#if KS_DEVICE && defined(KS_CLAIM_STATE_AGENT_SCOPE)
    uint64_t* s = (uint64_t*)&p[c];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FastClaim) / 8); ++i) __hip_atomic_store(&s[i], o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    u64_alias* s = (u64_alias*)&p[c];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FastClaim) / 8); ++i) s[i] = o[i];
#endif
  }
};
template <bool GS> KS_FN ClaimStates<GS> fast_uniform(ClaimStates<GS> c) { c.p = fast_uniform(c.p); return c; }
// The plan the engine is compiled for (FastPlan::global_state): 0 everything in LDS, 1 claim state in HBM, 2 claim state and
// order arrays in HBM. The order's accesses are plain loads and stores: the wavefront is the only reader and writer, its vector
// memory operations execute in order, and W::sync() (a wavefront-scope fence) stands between a store and another lane's load.
template <int GS> struct FastMem {
  static constexpr bool kStateHbm = GS >= 1, kOrderHbm = GS >= 2;
#if KS_DEVICE
  typedef typename std::conditional<kOrderHbm, uint16_t*, KS_LDS uint16_t*>::type o16;
#else
  typedef uint16_t* o16;
#endif
  typedef ClaimStates<kStateHbm> States;
};

// What ksolve_pack_fast reads its problem from: ONE record in HBM (not kernel arguments: a by-value argument whose address
// is taken is copied to private memory, and loads from private memory are divergent to the compiler — every branch of the
// wave-uniform loop would be compiled as a divergent one).
struct FastArgs { ProblemView pv; Workspace ws; FastWork fw; };

// a wave-uniform value the compiler cannot prove uniform: pin it to scalar registers
KS_FN int fast_uniform(int v) {
#if KS_DEVICE
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}
#if KS_DEVICE
template <class T>
KS_FN KS_LDS T* fast_uniform(KS_LDS T* p) {
  return (KS_LDS T*)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
}
#endif
template <class T>
KS_FN T* fast_uniform(T* p) {
#if KS_DEVICE
  const uint64_t u = (uint64_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return (T*)((uint64_t)lo | ((uint64_t)hi << 32));
#else
  return p;
#endif
}

#define KS_LIKELY(x) __builtin_expect(!!(x), 1)
#define KS_UNLIKELY(x) __builtin_expect(!!(x), 0)
#if KS_DEVICE
#define KS_COLD __device__ __attribute__((noinline))
#else
#define KS_COLD inline
#endif

// requirement-set cache helpers (per lane)
KS_FN uint32_t fast_hash(uint64_t vm) { return (uint32_t)((vm * 0x9E3779B97F4A7C15ull) >> 54) & (kFastEnt - 1); }   // the TOP bits of the product: the only ones every input bit reaches (the template id sits in bits 56..63)
// entry of requirement set vm (copied to `out`), or -1 (not cached yet); one 32-byte LDS read per probe
KS_FN int fast_lookup(const KS_LDS FastEnt* ent, uint64_t vm, FastEnt& out) {
  uint32_t h = fast_hash(vm);
  for (int probe = 0; probe < kFastEnt; ++probe) {
    out = lds_get(&ent[h]);
    if (!(out.info & 1u)) return -1;
    if (out.vmask == vm) return (int)h;
    h = (h + 1) & (kFastEnt - 1);
  }
  return -1;
}
KS_FN bool fast_fits_first(const FastEnt& e, const int32_t* req, const int32_t* size) {
  // bitwise on purpose: four compares and three ANDs, no short-circuit branches
  return (int)(size[0] <= e.cap[0] - req[0]) & (int)(size[1] <= e.cap[1] - req[1]) & (int)(size[2] <= e.cap[2] - req[2]) & (int)(size[3] <= e.cap[3] - req[3]);
}
// "some instance type of the entry holds `req` + `size`" — CanAdd's filterInstanceTypesByRequirements verdict
KS_FN bool fast_fits(const KS_LDS int32_t* pool, const FastEnt& e, const int32_t* req, const int32_t* size) {
  if (fast_fits_first(e, req, size)) return true;
  const int extra = (int)((e.info >> 8) & 0xFFu);
  const int off = (int)(e.info >> 16);
  for (int i = 0; i < extra; ++i) {
    bool o2 = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) o2 = o2 && size[r] <= pool[(off + i) * 4 + r] - req[r];
    if (o2) return true;
  }
  return false;
}
// Requirements.Compatible + Add on the packed form: every key the class selects on keeps at least one value. Adding the
// all-ones field to the field carries into the guard bit above it exactly when the field is not empty.
KS_FN bool fast_fields_ok(uint64_t m, uint64_t dmask) {
  const uint64_t g = (dmask << 1) & ~dmask;
  return (((m & dmask) + dmask) & g) == g;
}
KS_FN bool fast_sampled(int n, int p) {   // choosePivot's nine positions (pdq_emul.h sort()); n >= 50
  const unsigned q = (unsigned)n >> 2, u = (unsigned)p;
  return u - (q - 1) <= 2u || u - (2 * q - 1) <= 2u || u - (3 * q - 1) <= 2u;
}

// Everything that happens rarely (a new requirement set, a new claim, a new class slot, pdqsort leaving its single-move
// path): real function calls, so that their code and registers stay out of the loop that places a pod.
template <class W, int GS = 0>
struct FastCold {
  const ProblemView* Pk;
  const Workspace* Sk;
  const FastWork* Fk;
  typedef typename FastMem<GS>::o16 o16;
  ClaimOrder<W, o16, false> order;
  o16 snap;                 // [cap] order snapshot around a slow sort
  typename FastMem<GS>::States cst;
  KS_LDS FastEnt* ent;
  KS_LDS int32_t* pool;     // [kFastPool][4]
  KS_LDS FastSlot* aslot;
  KS_LDS FastMisc* Mp;
  int n_claims = 0, nv = 0, n_ent = 0, n_pool = 0, n_active = 0, n_evict = 0;
  uint32_t host_seq = 0, active_templates = 0;
  int bail_code = 0;
  int lo_ = 0, hi_ = -1;    // positions a slow sort permuted
  unsigned long long n_ref_extra = 0;

  KS_DEV void init(const ProblemView* p, const Workspace* s, const FastWork* f, char* lds) {
    Pk = p; Sk = s; Fk = f;
    const FastPlan& pl = f->plan;
    Mp = (KS_LDS FastMisc*)(lds + pl.off_misc);
    if constexpr (GS >= 1) cst.p = f->c_state; else cst.p = (KS_LDS FastClaim*)(lds + pl.off_state);
    if constexpr (GS >= 2) { order.key = (o16)f->o_key; order.ord = (o16)f->o_ord; snap = (o16)f->o_snap; }
    else { order.key = (o16)(lds + pl.off_key); order.ord = (o16)(lds + pl.off_ord); snap = (o16)(lds + pl.off_snap); }
    order.pos = nullptr;
    ent = (KS_LDS FastEnt*)(lds + pl.off_ent);
    pool = (KS_LDS int32_t*)(lds + pl.off_pool);
    aslot = (KS_LDS FastSlot*)(lds + pl.off_slot);
  }

  // F(requirement set vm) and its Pareto-maximal allocatable vectors -> a new cache entry
  KS_COLD int create_entry(uint64_t vm) {
    vm = W::uniform(vm);
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const int t = (int)(vm >> 56);
    const int iw = P.it_words, n_its = P.n_its, nr = P.n_res;
    const Dict& d = P.dict;
    const ProblemView& Pv = P;
    KS_LDS FastMisc& Mm = *Mp;
    const uint64_t* tits = S.t_its + (size_t)t * iw;
    KS_LDS uint64_t* its = Mp->its;
    const int nvv = nv;
    // compatible(it, reqs) (nodeclaim.go:620-622) for the keys pods select on; the template's other keys are in t_its already
    W::for_n(iw, [&](int w) {
      uint64_t acc = tits[w];
      for (int j = 0; j < nvv; ++j) {
        const int k = Mm.vkey[j];
        if (!((Pv.it_keys >> k) & 1u)) continue;
        if ((vm >> (Mm.voff[j] + Mm.vwidth[j])) & 1) continue;   // guard bit set: the requirement set does not define the key
        uint64_t field = (vm >> Mm.voff[j]) & (Mm.fmask[j] >> Mm.voff[j]);
        uint64_t r = Pv.key_undef[(size_t)k * iw + w];
        const size_t base = (size_t)Mm.vword[j] * 64;
        while (field) { const int b = ctz64(field); field &= field - 1; r |= Pv.kv_has[(base + b) * iw + w]; }
        acc &= r;
      }
      its[w] = acc;
    });
    // a compatible available offering (nodeclaim.go:624-638, types.go:553-570)
    uint32_t zones = (1u << P.n_zones) - 1, cts = (1u << P.n_cts) - 1;
    {
      const uint64_t* tm = P.tmpl_reqs.mask + (size_t)t * d.req_words;
      const uint32_t tdef = Mp->tdef[t];
      if (d.key_zone >= 0 && ((tdef >> d.key_zone) & 1u)) zones &= (uint32_t)tm[d.key_word_off[d.key_zone]];
      if (d.key_ct >= 0 && ((tdef >> d.key_ct) & 1u)) cts &= (uint32_t)tm[d.key_word_off[d.key_ct]];
      for (int j = 0; j < nv; ++j) {
        const uint32_t field = (uint32_t)((vm >> Mp->voff[j]) & (Mp->fmask[j] >> Mp->voff[j]));
        if (Mp->vkey[j] == d.key_zone) zones &= field;
        if (Mp->vkey[j] == d.key_ct) cts &= field;
      }
    }
    uint64_t cells = 0;
    for (uint32_t z = zones; z; z &= z - 1) cells |= (uint64_t)cts << (__builtin_ctz(z) * 4);
    for (int j = 0; j < iw; ++j) {
      const uint64_t in = its[j];
      const uint64_t okm = in ? W::ballot([&](int l) { const int it = j * 64 + l; return it < n_its && ((in >> l) & 1) && (Pv.it_off_avail[it] & cells) != 0; }) : 0ull;
      W::store(&its[j], okm);
    }
    W::sync();
    // slot
    if (n_ent * 5 >= kFastEnt * 4) return -1;
    uint32_t h = fast_hash(vm);
    while (ent[h].info & 1u) h = (h + 1) & (kFastEnt - 1);
    uint64_t* eits = F.ent_its + (size_t)h * iw;
    W::for_n(iw, [&](int w) { eits[w] = its[w]; });
    // Pareto-maximal allocatable vectors of F: repeatedly take the lexicographic maximum, drop what it dominates
    KS_LDS uint64_t* rem = Mp->rem; KS_LDS uint64_t* cand = Mp->cand;
    W::for_n(iw, [&](int w) { rem[w] = its[w]; });
    int count = 0;
    const int poff = n_pool;
    int32_t f0 = -1, f1 = -1, f2 = -1, f3 = -1;
    for (;;) {
      const uint64_t any = W::reduce_or(iw, [&](int w) { return (uint64_t)rem[w]; });
      if (!any) break;
      W::for_n(iw, [&](int w) { cand[w] = rem[w]; });
      int64_t v0 = 0x3FFFFFFF, v1 = 0x3FFFFFFF, v2 = 0x3FFFFFFF, v3 = 0x3FFFFFFF;
      for (int r = 0; r < nr; ++r) {
        const int64_t* al = P.it_alloc + (size_t)r * n_its;
        const int64_t mx = W::reduce_max_i64(iw * 64, [&](int it) { return (it < n_its && ((cand[it >> 6] >> (it & 63)) & 1)) ? al[it] : INT64_MIN; });
        if (r == 0) v0 = mx; else if (r == 1) v1 = mx; else if (r == 2) v2 = mx; else v3 = mx;
        for (int j = 0; j < iw; ++j) {
          const uint64_t in = cand[j];
          const uint64_t eq = in ? W::ballot([&](int l) { const int it = j * 64 + l; return it < n_its && ((in >> l) & 1) && al[it] == mx; }) : 0ull;
          W::store(&cand[j], eq);
        }
        W::sync();
      }
      if (count == 0) { f0 = (int32_t)v0; f1 = (int32_t)v1; f2 = (int32_t)v2; f3 = (int32_t)v3; }
      else {
        if (count > kFastMaxPareto || n_pool >= kFastPool) return -1;
        if (W::leader()) { pool[n_pool * 4 + 0] = (int32_t)v0; pool[n_pool * 4 + 1] = (int32_t)v1; pool[n_pool * 4 + 2] = (int32_t)v2; pool[n_pool * 4 + 3] = (int32_t)v3; }
        n_pool++;
      }
      count++;
      for (int j = 0; j < iw; ++j) {
        const uint64_t in = rem[j];
        const uint64_t dom = in ? W::ballot([&](int l) {
          const int it = j * 64 + l;
          if (it >= n_its || !((in >> l) & 1)) return false;
          bool le = Pv.it_alloc[it] <= v0;
          if (nr > 1) le = le && Pv.it_alloc[(size_t)n_its + it] <= v1;
          if (nr > 2) le = le && Pv.it_alloc[(size_t)2 * n_its + it] <= v2;
          if (nr > 3) le = le && Pv.it_alloc[(size_t)3 * n_its + it] <= v3;
          return le;
        }) : 0ull;
        W::store(&rem[j], in & ~dom);
      }
      W::sync();
    }
    if (W::leader()) {
      FastEnt e;
      e.vmask = vm; e.cap[0] = f0; e.cap[1] = f1; e.cap[2] = f2; e.cap[3] = f3;
      e.info = 1u | ((uint32_t)(count > 1 ? count - 1 : 0) << 8) | ((uint32_t)poff << 16);
      e.pad = 0;
      lds_put(&ent[h], e);
    }
    n_ent++;
    W::sync();
    return (int)h;
  }

  // Returns 0 when the problem is of the shape this engine solves, a reason code otherwise.
  KS_COLD int setup() {
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const Dict& d = P.dict;
    const int nk = d.n_keys, iw = P.it_words, nr = P.n_res, n_its = P.n_its, nc = P.n_classes, T = P.n_templates;
    const ProblemView& Pv = P;
    if (!P.plain || P.n_rows != P.n_pods || nr > 4 || T > 32 || nc > kFastClasses || iw > kMaxItWords) return 1;
    // (instance types may use any operator: with positive sets on the claim side the NotIn / DoesNotExist escape of
    // requirements.go:260-265 never applies, so compatible() stays monotone)
    // templates: only In sets
    if (W::reduce_or(T, [&](int t) { return (uint64_t)(Pv.tmpl_reqs.complement[t] | (Pv.tmpl_reqs.has_gte ? Pv.tmpl_reqs.has_gte[t] : 0) | (Pv.tmpl_reqs.has_lte ? Pv.tmpl_reqs.has_lte[t] : 0)); })) return 3;
    // classes: only In sets; the keys they define are the variable keys
    if (W::reduce_or(nc, [&](int c) { return (uint64_t)Pv.cls_reqs.complement[c]; })) return 4;
    const uint32_t vk = (uint32_t)W::reduce_or(nc, [&](int c) { return (uint64_t)Pv.cls_reqs.defined[c]; });
    if (d.key_hostname >= 0 && ((vk >> d.key_hostname) & 1u)) return 5;
    if (d.key_it >= 0 && ((vk >> d.key_it) & 1u)) return 5;
    int bits = 0;
    nv = 0;
    for (int k = 0; k < nk; ++k) {
      if (!((vk >> k) & 1u)) continue;
      if (d.key_word_off[k + 1] - d.key_word_off[k] != 1 || nv >= kFastMaxVar) return 6;
      const uint64_t valid = d.value_valid[d.key_word_off[k]];
      const int width = valid ? 64 - __builtin_clzll(valid) : 1;
      if (bits + width + 1 > kFastVarBits) return 6;
      if (W::leader()) {
        Mp->vkey[nv] = (uint8_t)k; Mp->voff[nv] = (uint8_t)bits; Mp->vwidth[nv] = (uint8_t)width; Mp->vword[nv] = (uint16_t)d.key_word_off[k];
        Mp->fmask[nv] = ((1ull << width) - 1) << bits;
      }
      bits += width + 1;   // + the guard bit
      nv++;
    }
    W::sync();
    if (W::leader()) {
      FastVar fv;
      fv.nv = nv;
      for (int j = 0; j < kFastMaxVar; ++j) { fv.vkey[j] = Mp->vkey[j]; fv.voff[j] = Mp->voff[j]; fv.vwidth[j] = Mp->vwidth[j]; fv.vword[j] = Mp->vword[j]; }
      *F.var = fv;
    }
    W::sync();
    // every quantity in 31 bits
    if (W::reduce_or(nr * n_its, [&](int i) { const int64_t a = Pv.it_alloc[i]; return (uint64_t)((a >= (1ll << 30) || a <= -(1ll << 30)) ? 1 : 0); })) return 7;
    if (W::reduce_or(nr * nc, [&](int i) { const int64_t a = Pv.cls_requests[i]; return (uint64_t)((a >= (1ll << 30) || a < 0) ? 1 : 0); })) return 7;
    // templates: packed form, and NewScheduler's prefilter (scheduler.go:156-171) with positive sets
    KS_LDS FastMisc& Mm = *Mp;
    const int nvv = nv;
    W::for_n(T, [&](int t) {
      const uint64_t* tm = Pv.tmpl_reqs.mask + (size_t)t * d.req_words;
      const uint32_t tdef = Pv.tmpl_reqs.defined[t];
      uint64_t vm = (uint64_t)t << 56;
      for (int j = 0; j < nvv; ++j) {
        if ((tdef >> Mm.vkey[j]) & 1u) vm |= (tm[Mm.vword[j]] << Mm.voff[j]) & Mm.fmask[j];
        else vm |= Mm.fmask[j] | (1ull << (Mm.voff[j] + Mm.vwidth[j]));   // undefined: every value, and the guard bit says so
      }
      Mm.tvmask[t] = vm; Mm.tdef[t] = tdef;
    });
    active_templates = 0;
    for (int t = 0; t < T; ++t) {
      const uint64_t* tm = P.tmpl_reqs.mask + (size_t)t * d.req_words;
      const uint32_t tdef = Mp->tdef[t];
      uint64_t* tits = S.t_its + (size_t)t * iw;
      const uint64_t* tin = P.tmpl_its + (size_t)t * iw;
      W::for_n(iw, [&](int w) {
        uint64_t acc = tin[w] & Pv.it_alloc_ok[w];
        for (int k = 0; k < nk; ++k) {
          if (!((tdef >> k) & 1u)) continue;
          const uint32_t w0 = d.key_word_off[k], w1 = d.key_word_off[k + 1];
          if (k == d.key_it) { acc &= tm[w0 + w]; continue; }
          if (!((Pv.it_keys >> k) & 1u)) continue;
          uint64_t r = Pv.key_undef[(size_t)k * iw + w];
          for (uint32_t x = w0; x < w1; ++x) for (uint64_t b = tm[x]; b; b &= b - 1) r |= Pv.kv_has[((size_t)x * 64 + ctz64(b)) * iw + w];
          acc &= r;
        }
        tits[w] = acc;
      });
      uint32_t zones = (1u << P.n_zones) - 1, cts = (1u << P.n_cts) - 1;
      if (d.key_zone >= 0 && ((tdef >> d.key_zone) & 1u)) zones &= (uint32_t)tm[d.key_word_off[d.key_zone]];
      if (d.key_ct >= 0 && ((tdef >> d.key_ct) & 1u)) cts &= (uint32_t)tm[d.key_word_off[d.key_ct]];
      uint64_t cells = 0;
      for (uint32_t z = zones; z; z &= z - 1) cells |= (uint64_t)cts << (__builtin_ctz(z) * 4);
      uint64_t any = 0;
      for (int j = 0; j < iw; ++j) {
        const uint64_t in = tits[j];
        const uint64_t okm = in ? W::ballot([&](int l) { const int it = j * 64 + l; return it < n_its && ((in >> l) & 1) && (Pv.it_off_avail[it] & cells) != 0; }) : 0ull;
        W::store(&tits[j], okm);
        any |= okm;
      }
      W::sync();
      if (any) active_templates |= 1u << t;
      int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
      const int64_t* lim = P.tmpl_limits + (size_t)t * (nr + 1);
      W::for_n(nr + 1, [&](int r) { rem[r] = lim[r]; });
    }
    // classes: packed form + the (class, template) verdicts that never change: taints (nodeclaim.go:126) and keys the
    // template does not define (requirements.go:185-193; with positive operators such a key stays undefined for good)
    FastSlot* fc = F.cls;
    KS_LDS uint16_t* so = Mp->slot_of;
    const uint32_t wk = d.well_known_mask;
    const uint64_t bad = W::reduce_or(nc, [&](int c) {
      const uint64_t* cm = Pv.cls_reqs.mask + (size_t)c * d.req_words;
      const uint32_t kdef = Pv.cls_reqs.defined[c];
      uint64_t vm = 0xFFull << 56, dm = 0, badc = 0;
      for (int j = 0; j < nvv; ++j) {
        if ((kdef >> Mm.vkey[j]) & 1u) {
          const uint64_t f = (cm[Mm.vword[j]] << Mm.voff[j]) & Mm.fmask[j];
          if (!f) badc = 1;   // In [] == DoesNotExist: not positive
          vm |= f; dm |= Mm.fmask[j];
        } else vm |= Mm.fmask[j] | (1ull << (Mm.voff[j] + Mm.vwidth[j]));
      }
      FastSlot s;
      s.cvmask = vm; s.dmask = dm;
      for (int r = 0; r < 4; ++r) s.size[r] = r < nr ? (int32_t)Pv.cls_requests[(size_t)c * nr + r] : 0;
      uint32_t ok = 0;
      const uint64_t tol = Pv.cls_tolerates[c];
      for (int t = 0; t < T; ++t) if (!(Pv.tmpl_taints[t] & ~tol) && !(kdef & ~Mm.tdef[t] & ~wk)) ok |= 1u << t;
      s.tmplok = ok; s.kdef = kdef;
      fc[c] = s;
      so[c] = 0xFFFF;
      return badc;
    });
    if (bad) return 8;
    W::for_n(kFastEnt, [&](int i) { ent[i].info = 0; });
    return 0;
  }

  // a class seen for the first time (or after an eviction) gets a slot; bit 16 of the result: every cursor must restart
  KS_COLD int new_slot(int k) {
    k = (int)W::uniform((uint64_t)(uint32_t)k);
    KS_LDS uint16_t* so = Mp->slot_of;
    int evicted = 0;
    if (n_active == kFastSlots) {
      // every slot taken: forget them all (their classes start again from position 0 if they ever come back)
      KS_LDS uint16_t* act = Mp->active;
      W::for_n(kFastSlots, [&](int i) { so[act[i]] = 0xFFFF; });
      n_active = 0;
      n_evict++;
      evicted = 1 << 16;
    }
    const int slot = n_active++;
    const FastSlot rec = Fk->cls[k];
    if (W::leader()) { lds_put(&aslot[slot], rec); Mp->active[slot] = (uint16_t)k; so[k] = (uint16_t)slot; }
    W::sync();
    return slot | evicted;
  }

  // any path of pdqsort other than the single stable move: run the emulation, compare the order before and after
  // (lo_..hi_ = the positions whose claim changed, hi_ < lo_: none)
  KS_COLD void slow_sort(int n, int defect, int app) {
    order.n = (int)W::uniform((uint64_t)(uint32_t)n); order.defect = (int)W::uniform((uint64_t)(uint32_t)defect); order.defect_append = W::uniform((uint64_t)app) != 0;
    n = order.n;
    const o16 oo = order.ord; const o16 sn = snap;
    if constexpr (FastMem<GS>::kOrderHbm) W::copy8(sn, oo, n); else W::for_n(n, [&](int i) { sn[i] = oo[i]; });
    order.sort();
    lo_ = order.ff(0, n, [&](int i) { return sn[i] != oo[i]; });
    if (lo_ >= n) { lo_ = 0; hi_ = -1; return; }
    hi_ = order.fl(0, n, [&](int i) { return sn[i] != oo[i]; });
  }
  // The new claim (appended at n-1 with one pod) takes its place behind the last claim with at most one pod. Returns its
  // position b >= 0 (Mp->acc = the class slots that accept it), -1 when pdqsort left the single-move path (lo_/hi_), -2: stop.
  KS_COLD int place_new_claim(int n) {
    n = (int)W::uniform((uint64_t)(uint32_t)n);
    const int a = n - 1;
    const int moved = (int)order.ord[a];
    const bool exact = n <= 12 || (n >= 50 && !fast_sampled(n, a));
    if (!exact) { slow_sort(n, a, 1); return -1; }
    order.n = n; order.defect = a; order.defect_append = true;
    order.sort();
    const o16 kq = order.key;
    const int b = W::find_first(0, n, [&](int i) { return kq[i] > 1u; }) - 1;   // the claims with one pod are the prefix it joined the end of
    const int nac = n_active;
    const FastClaim nst = cst.get((uint32_t)moved);
    const int t = (int)(nst.vmask >> 56);
    for (int j = 0; j < kFastRows; ++j) {
      uint64_t accm = 0;
      uint64_t todo = j * 64 < nac ? W::ballot([&](int l) { return j * 64 + l < nac; }) : 0ull;
      while (todo) {
        LaneVar<uint64_t> missv;
        const uint64_t td = todo;
        uint64_t okb = 0, miss = 0, d0, d1;
        W::ballot4([&](int l) {
          missv.at(l) = 0;
          if (!((td >> l) & 1)) return 0;
          const FastSlot s = lds_get(&aslot[j * 64 + l]);
          if (!((s.tmplok >> t) & 1u)) return 0;
          const uint64_t m = nst.vmask & s.cvmask;
          if (!fast_fields_ok(m, s.dmask)) return 0;
          FastEnt e;
          if (fast_lookup(ent, m, e) < 0) { missv.at(l) = m; return 2; }
          return fast_fits(pool, e, nst.req, s.size) ? 1 : 0;
        }, okb, miss, d0, d1);
        accm |= okb;
        todo = miss;
        if (miss && create_entry(missv.bcast(ctz64(miss))) < 0) { bail_code = 20; return -2; }
      }
      W::store(&Mp->acc[j], accm);
    }
    W::sync();
    return b;
  }

  // addToNewNodeClaim (scheduler.go:695-790) for a pod no in-flight claim accepted: 1 = claim n created (appended to the
  // order with one pod), 0 = the engine must stop (bail_code; -1 = capacity).
  KS_COLD int new_claim(int slot, int bi, int n) {
    slot = (int)W::uniform((uint64_t)(uint32_t)slot); bi = (int)W::uniform((uint64_t)(uint32_t)bi); n = (int)W::uniform((uint64_t)(uint32_t)n);
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const FastSlot cs = lds_get(&aslot[slot]);
    const int T = P.n_templates, nr = P.n_res, iw = P.it_words, cap = F.plan.cap;
    n_ref_extra += (unsigned long long)n;
    for (int t = 0; t < T; ++t) {
      if (!((active_templates >> t) & 1u)) continue;
      const uint32_t lm = P.tmpl_limit_mask[t];
      if (lm) {
        // filterByRemainingResources (scheduler.go:1069-1085): this engine only continues while no type is excluded
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        if (((lm >> nr) & 1) && rem[nr] <= 0) { bail_code = 23; return 0; }
        const ProblemView& Pv = P;
        const uint64_t* tits = S.t_its + (size_t)t * iw;
        const int n_its = P.n_its;
        uint64_t excluded = 0;
        for (int w = 0; w < iw; ++w) {
          const uint64_t in = tits[w];
          if (!in) continue;
          excluded |= W::ballot([&](int l) {
            const int it = w * 64 + l;
            if (it >= n_its || !((in >> l) & 1)) return false;
            bool v = true;
            for (int q = 0; q < nr; ++q) if ((lm >> q) & 1) v = v && Pv.it_cap[(size_t)q * n_its + it] <= rem[q];
            return !v;
          });
        }
        if (excluded) { bail_code = 24; return 0; }
      }
      host_seq++;
      n_ref_extra++;
      if (!((cs.tmplok >> t) & 1u)) continue;
      const uint64_t m = Mp->tvmask[t] & cs.cvmask;
      if (!fast_fields_ok(m, cs.dmask)) continue;
      FastEnt e;
      int eh = fast_lookup(ent, m, e);
      if (eh < 0) { eh = create_entry(m); if (eh < 0) { bail_code = 25; return 0; } e = lds_get(&ent[eh]); }
      const int32_t zero[4] = {0, 0, 0, 0};
      if (!fast_fits(pool, e, zero, cs.size)) continue;
      if (n_claims >= S.max_claims) { bail_code = -1; return 0; }   // capacity: reported as such
      if (n_claims >= cap) { bail_code = 26; return 0; }
      const int c = n_claims++;
      if (W::leader()) {
        FastClaim ns;
        ns.vmask = m;
        for (int q = 0; q < 4; ++q) ns.req[q] = cs.size[q];
        cst.put((uint32_t)c, ns);
        F.c_hostseq[c] = host_seq;
        order.key[n] = 1; order.ord[n] = (uint16_t)c;   // order.append
      }
      W::sync();
      if (lm) {
        // subtractMax (scheduler.go:1049-1066) over the claim's instance types: F(m) ∩ fits(size)
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        const uint64_t* eits = F.ent_its + (size_t)eh * iw;
        const ProblemView& Pv = P;
        const int n_its = P.n_its;
        for (int q = 0; q < nr; ++q) if ((lm >> q) & 1) {
          const int64_t mx = W::reduce_max_i64(n_its, [&](int it) {
            if (!((eits[it >> 6] >> (it & 63)) & 1)) return INT64_MIN;
            for (int z = 0; z < nr; ++z) if (Pv.it_alloc[(size_t)z * n_its + it] < (int64_t)cs.size[z]) return INT64_MIN;
            return Pv.it_cap[(size_t)q * n_its + it];
          });
          W::store(&rem[q], rem[q] - mx);
        }
        W::sync();
      }
      return 1;
    }
    bail_code = 27;   // an unschedulable pod: error codes and diagnostics come from the general engine
    return 0;
  }

  KS_COLD void finish(int status, int n, unsigned long long steps, unsigned long long n_steps, unsigned long long n_tests, unsigned long long n_ref, const unsigned long long* tc) {
    // results: the final order (the defect of the last commit stays unsorted, as in the reference), the claims' state and
    // the cache entry of each claim's requirement set
    const Workspace& S = *Sk; const FastWork& F = *Fk;
    if (status != 3 && status != 1) {
      uint32_t* go = S.o_ord;
      const o16 oo = order.ord; const o16 ok_ = order.key;
      FastClaim* gs = F.c_state; uint32_t* gn = F.c_npods; uint16_t* ge = F.c_ent;
      const typename FastMem<GS>::States ls = cst;
      const KS_LDS FastEnt* en = ent;
      W::for_n(n, [&](int i) { const uint32_t c = oo[i]; go[i] = c; gn[c] = ok_[i]; });
      W::for_n(n, [&](int c) {
        const FastClaim st = ls.get((uint32_t)c);
        FastEnt e;
        if constexpr (GS == 0) gs[c] = st;   // otherwise the state has lived in F.c_state all along
        ge[c] = (uint16_t)fast_lookup(en, st.vmask, e);
      });
      W::store(S.n_claims_out, n_claims);
    }
    if (status) W::store(S.status_out, status);
    Counters c{};
    c.bin_evaluations = n_tests; c.full_evaluations = n_steps; c.queue_pops = steps; c.sorts = steps; c.slow_sorts = order.slow_sorts;
    c.column_resets = (unsigned long long)n_evict; c.ref_bin_evaluations = n_ref + n_ref_extra;
    c.cycles[20] = (unsigned long long)(bail_code > 0 ? bail_code : 0);
    if (tc) for (int i = 0; i < 16; ++i) c.cycles[i] = tc[i];
    if (W::leader()) *S.counters = c;
    W::sync();
  }
};

// The loop's state between two events (LDS): scalars, the cursors and the 64-entry queue block, one value per lane
struct FastHot {
  int base, bi, bn, n, np, max_steps, steps, status;
  int pend_a, pend_x, pend_new, ev_arg;
  uint32_t pend_mv, pad0;
  uint64_t ev_vm;
  unsigned long long n_steps, n_tests, n_ref, hot_cycles;
  unsigned long long tsec[8];   // profiling builds: shader clock per path of the loop
  const uint32_t* q_class; const volatile int* cancel; uint32_t* q_claim; uint32_t* q_cnt;
  uint32_t cur[kFastRows][64];
  uint32_t bcls[64], bslot[64], oclaim[64], ocnt[64], nxt_cls[64];
};
template <int GS>
struct FastHotCtx {   // LDS pointers of the loop, passed by value
  typename FastMem<GS>::o16 okey, oord; typename FastMem<GS>::States cst; KS_LDS FastEnt* ent; KS_LDS int32_t* pool;
  KS_LDS FastSlot* aslot; KS_LDS uint16_t* slot_of; KS_LDS FastHot* hs;
};
enum { FEV_DONE = 0, FEV_ENTRY = 1, FEV_SLOT = 2, FEV_SLOWSORT = 3, FEV_PLACE = 4, FEV_NEWCLAIM = 5, FEV_COUNT = 6 };

// The loop that places pods: a function of its own, WITHOUT calls — whatever happens rarely (a requirement set seen for the
// first time, a new class slot, a new claim, pdqsort leaving its single-move path) ends the run with an event code; the
// driver handles it through FastCold and runs the loop again. So the compiler allocates registers for this loop alone.
template <class W, int GS>
KS_COLD int fast_hot_run(FastHotCtx<GS> cx) {
  typedef typename FastMem<GS>::o16 o16;
  const unsigned long long t_in = W::clock();
  const o16 okey = fast_uniform(cx.okey), oord = fast_uniform(cx.oord);
  const typename FastMem<GS>::States cst = fast_uniform(cx.cst);
  KS_LDS FastEnt* const ent = fast_uniform(cx.ent);
  KS_LDS int32_t* const pool = fast_uniform(cx.pool);
  KS_LDS FastSlot* const aslot = fast_uniform(cx.aslot);
  KS_LDS uint16_t* const slot_of = fast_uniform(cx.slot_of);
  KS_LDS FastHot* const hs = fast_uniform(cx.hs);
  // ---- state in ----
  const int np = fast_uniform(hs->np), max_steps = fast_uniform(hs->max_steps);
  const KS_GLOBAL uint32_t* const gqcls = (const KS_GLOBAL uint32_t*)fast_uniform(hs->q_class);
  const volatile int* const cancel = fast_uniform(hs->cancel);
  KS_GLOBAL uint32_t* const gqclaim = (KS_GLOBAL uint32_t*)fast_uniform(hs->q_claim);
  KS_GLOBAL uint32_t* const gqcnt = (KS_GLOBAL uint32_t*)fast_uniform(hs->q_cnt);
  int base = fast_uniform(hs->base), bi = fast_uniform(hs->bi), bn = fast_uniform(hs->bn), n = fast_uniform(hs->n), steps = fast_uniform(hs->steps), status = fast_uniform(hs->status);
  int pend_a = fast_uniform(hs->pend_a), pend_x = fast_uniform(hs->pend_x); uint32_t pend_mv = (uint32_t)fast_uniform((int)hs->pend_mv); bool pend_new = fast_uniform(hs->pend_new) != 0;
  unsigned long long n_steps = W::uniform(hs->n_steps), n_tests = W::uniform(hs->n_tests), n_ref = W::uniform(hs->n_ref);
  LaneVar<uint32_t> cur[kFastRows], nxt_cls, bcls, bslot, oclaim, ocnt;
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < kFastRows; ++j) cur[j].at(l) = hs->cur[j][l];
    nxt_cls.at(l) = hs->nxt_cls[l]; bcls.at(l) = hs->bcls[l];
    bslot.at(l) = hs->bslot[l]; oclaim.at(l) = hs->oclaim[l]; ocnt.at(l) = hs->ocnt[l];
  });
  int ev = FEV_DONE, ev_arg = 0; uint64_t ev_vm = 0;
  // ---- group speculation ----
  // One CanAdd test serves up to eight queue entries: lane 8j+q tests queue entry bi+j (its class) against the claim at
  // position cursor(class)+q — the same three dependent LDS reads as a window test of one pod. The verdicts are taken
  // against the state at that moment; they are used for the entries one after the other, in queue order:
  //   * a rejection is final (fact 3 above), whatever happened to the claim since;
  //   * an acceptance holds while the claim has not gained a pod since the test (g_touched: lanes whose claim has);
  //   * claims only move right, past claims with fewer pods, so between an entry's cursor and its first accepting lane
  //     there are only claims that rejected it (those of its window, those that came from the left of its cursor); a
  //     claim from the right never gets in front of it. So entry bi+j goes to the claim of its first accepting lane iff
  //     that claim is untouched — addToInflightNode's "lowest index that accepts" (scheduler.go:667-686) without a new
  //     test. gp follows every lane's claim through the moves (the same update as the cursors).
  // Every other entry (first acceptor touched, an unresolved lane before it, nothing in eight positions) takes the window
  // test below; the rest of the group stays valid behind it. A move that is not one short shift (the pending path,
  // pdqsort's other paths, a new claim) drops the group.
  LaneVar<uint32_t> gx, gk, gp, gB, gq0, gq1, gq2, gq3;   // gB: a lower bound of the position of the first claim behind the entry's eight
  LaneVar<uint64_t> gm;
  uint64_t g_acc = 0, g_odd = 0, g_touched = 0, g_jumped = 0;   // g_jumped: lanes whose claim a commit moved past another claim
  int gj = 0, gn = 0;   // entries gj .. gn-1 of the group are still to be placed; entry gj is queue entry bi
  // choosePivot's sampled positions (fast_sampled) as three starts; n is fixed inside one run of this function. With
  // 12 < n < 50 every re-sort that has something to move is pdqsort's other path (mid_n). A commit that leaves the order
  // sorted as it stands — the next claim has at least the new count — needs no sort at all, whatever n and the position:
  // pdqsort finds no descent and does nothing.
  const bool use_groups = max_steps < 0;
  const bool mid_n = n > 12 && n < 50;
  const uint32_t e1 = n >= 50 ? (uint32_t)(n >> 2) - 1u : 0x7FFFFFF0u, e2 = n >= 50 ? 2u * (uint32_t)(n >> 2) - 1u : 0x7FFFFFF0u, e3 = n >= 50 ? 3u * (uint32_t)(n >> 2) - 1u : 0x7FFFFFF0u;
#ifdef KSOLVE_PHASE_TIMERS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0, ts7 = 0, tlast = W::clock();
#define KS_SEC(v) { const unsigned long long t_ = W::clock(); v += t_ - tlast; tlast = t_; }
#else
#define KS_SEC(v)
#endif
  for (;;) {
    // ---- the next block of the queue ----
    if (KS_UNLIKELY(bi >= bn)) {
      if (bn > 0) {   // the finished block's results
        const int dn = bn, b0 = base;
        W::each([&](int l) { if (l < dn) { gqclaim[b0 + l] = oclaim.at(l); gqcnt[b0 + l] = ocnt.at(l); } });
        base += 64;
      }
      if (base >= np || status) { bn = 0; bi = 0; break; }
      bn = np - base < 64 ? np - base : 64;
      bi = 0; gj = 0; gn = 0;
      const int bnn = bn, nb = base + 64;
      W::each([&](int l) { bcls.at(l) = nxt_cls.at(l); bslot.at(l) = l < bnn ? (uint32_t)slot_of[nxt_cls.at(l)] : 0xFFFFu; });
      W::each([&](int l) { if (nb + l < np) nxt_cls.at(l) = gqcls[nb + l]; });
      if ((base & 1023) == 0 && cancel) {
        // > 0: ksolve_cancel / the deadline. < 0 (tests only, KSOLVE_TEST_CANCEL_AT): as if the cancel landed once -flag pods were placed
        const int cv = fast_uniform((int)W::poll_flag(cancel));
        if (cv > 0 || (cv < 0 && base >= -cv)) { status = 2; bn = 0; break; }
      }
    }
    if (KS_UNLIKELY(max_steps >= 0 && steps >= max_steps)) { status = 2; break; }
    // ---- sort.Slice (scheduler.go:598) for a move the last commit left behind ----
    if (KS_UNLIKELY(pend_a >= 0 || pend_new)) {
      if (pend_new) { ev = FEV_PLACE; break; }
      gj = 0; gn = 0;
      const int a = pend_a;
      if (!(a + 1 >= n || (uint32_t)okey[a + 1 < n ? a + 1 : a] >= pend_mv) &&   // something to move ...
          !(n <= 12 || (n >= 50 && !fast_sampled(n, a)))) { ev = FEV_SLOWSORT; ev_arg = a; break; }   // ... and not by the single stable move
      pend_a = -1;
      // one stable move: the claim at a (count pend_mv) goes right past the claims with a smaller count
      int from = a;
      const uint32_t mv = pend_mv;
      for (;;) {
        LaneVar<uint32_t> kv, ov;
        const uint64_t less = W::ballot([&](int l) {
          const int i = from + 1 + l;
          if (i >= n) return false;
          const uint32_t k = okey[i];
          kv.at(l) = k; ov.at(l) = oord[i];
          return k < mv;
        });
        const int s_ = less == ~0ull ? 64 : ctz64(~less);   // sorted beyond a: the smaller counts are a prefix
        if (s_ == 0) break;
        W::each([&](int l) { if (l < s_) { okey[from + l] = (uint16_t)kv.at(l); oord[from + l] = (uint16_t)ov.at(l); } });
        from += s_;
        if (s_ < 64) break;
      }
      if (from != a) {
        if (W::leader()) { okey[from] = (uint16_t)mv; oord[from] = (uint16_t)pend_x; }
        W::sync();
        const int b = from;   // positions (a, b] moved left by one
        W::each([&](int l) {
#pragma unroll
          for (int j = 0; j < kFastRows; ++j) { const uint32_t r = cur[j].at(l); cur[j].at(l) = r - (uint32_t)(((uint32_t)a < r && r <= (uint32_t)b) ? 1 : 0); }
        });
      }
    }
    KS_SEC(ts0)   // block fetch, pending move
    // ---- the pod's class slot ----
    const int slot = (int)bslot.bcast(bi);
    if (KS_UNLIKELY(slot == 0xFFFF)) { ev = FEV_SLOT; ev_arg = (int)bcls.bcast(bi); break; }
    // The queue's last entry is not placed from a group: no add follows it, its move stays undone (as in the reference). The
    // same holds for the last entry before a block boundary at which the cancel flag is polled: a cancelled Solve() ends there,
    // and the claim order it reports is the one of the last sort the reference would have run (scheduler.go:598 sorts at the
    // start of an add, never after the last one).
    const int lastq = (base + bn >= np || (cancel && ((base + 64) & 1023) == 0)) ? 1 : 0;
    if (use_groups && gj >= gn && bi + lastq < bn) {
      // ---- a new group: the next entries of the block that have a class slot, eight at most ----
      const int g0 = bn - bi - lastq < 8 ? bn - bi - lastq : 8, bi0 = bi, nn = n;
      LaneVar<uint32_t> gs, gr;
      W::each([&](int l) { const int j = l >> 3; gs.at(l) = bslot.shuffle(l, (bi0 + (j < g0 ? j : 0)) & 63); });
      W::each([&](int l) {
        const uint32_t sv = gs.at(l);
        const int sl = (int)(sv & 63u);
        const uint32_t a0 = cur[0].shuffle(l, sl), a1 = cur[1].shuffle(l, sl), a2 = cur[2].shuffle(l, sl), a3 = cur[3].shuffle(l, sl);
        const uint32_t row = (sv >> 6) & 3u;
        gr.at(l) = row == 0 ? a0 : row == 1 ? a1 : row == 2 ? a2 : a3;
      });
      const uint64_t nos = W::ballot([&](int l) { return (l >> 3) >= g0 || gs.at(l) == 0xFFFFu; });
      gn = nos ? ctz64(nos) >> 3 : 8;   // >= 1: this entry has its slot
      gj = 0; g_touched = 0; g_jumped = 0;
      const int gnn = gn;
      W::ballot2([&](int l) {
        const int j = l >> 3, q = l & 7;
        const int p = (int)gr.at(l) + q;
        gx.at(l) = 0xFFFFFFFFu; gk.at(l) = 0; gp.at(l) = 0xFFFFFFFFu; gm.at(l) = 0; gq0.at(l) = 0; gq1.at(l) = 0; gq2.at(l) = 0; gq3.at(l) = 0;
        gB.at(l) = gr.at(l) + 8u;
        if (j >= gnn || p >= nn) return 0;
        const uint32_t x = oord[p];
        const FastClaim st = cst.get(x);
        const FastSlot s = lds_get(&aslot[gs.at(l)]);
        const uint64_t m = st.vmask & s.cvmask;
        FastEnt e = lds_get(&ent[fast_hash(m)]);
        gx.at(l) = x; gk.at(l) = okey[p]; gp.at(l) = (uint32_t)p; gm.at(l) = m;
        gq0.at(l) = st.req[0] + s.size[0]; gq1.at(l) = st.req[1] + s.size[1]; gq2.at(l) = st.req[2] + s.size[2]; gq3.at(l) = st.req[3] + s.size[3];
        const int base_ok = (int)((s.tmplok >> (st.vmask >> 56)) & 1u) & (int)fast_fields_ok(m, s.dmask);
        int simple = (int)(e.info & 1u) & (int)(e.vmask == m);
        if (base_ok & (simple ^ 1) & (int)(e.info & 1u)) simple = (int)(fast_lookup(ent, m, e) >= 0);   // not the cache's first probe: the probe sequence
        int fit = (int)fast_fits_first(e, st.req, s.size);
        if (base_ok & simple & (fit ^ 1) & (int)(((e.info >> 8) & 0xFFu) != 0)) fit = (int)fast_fits(pool, e, st.req, s.size);   // the other Pareto vectors
        // bit 0: accepted; bit 1: the requirement set is not cached yet (the window test raises the event)
        return (base_ok & simple & fit) | ((base_ok & (simple ^ 1)) << 1);
      }, g_acc, g_odd);
      n_tests += (unsigned long long)(8 * gnn);
      n_steps++;
      KS_SEC(ts1)   // group test
#ifdef KSOLVE_PHASE_TIMERS
      ts5++;
#endif
    }
    // ---- entries of the group that take no window test ----
    bool placed_any = false;
    while (gj < gn) {
      const int j = gj;
      const uint32_t gb = (uint32_t)(g_acc >> (8 * j)) & 0xFFu, ob = (uint32_t)(g_odd >> (8 * j)) & 0xFFu;
      const int qf = __builtin_ctz(gb | 0x100u);
      if (qf == 8 || (ob & ((1u << qf) - 1u))) break;   // nothing in eight positions, or an unresolved lane first: the window test
      int L = 8 * j + qf;
      if (KS_UNLIKELY(((g_touched & g_jumped) >> L) & 1)) {
        // The first accepting lane's claim moved past other claims since the test: the entry's first candidate is now the
        // accepting lane whose claim stands leftmost. It may be used when everything between the class's cursor and it is
        // known to reject the entry: no unresolved lane before it, and it has not passed the first claim behind the eight.
        const uint64_t accj = g_acc & (0xFFull << (8 * j)), oddj = g_odd & (0xFFull << (8 * j));
        int c = -1;
        const uint32_t pmin = W::argmin_u32([&](int l) { return ((accj >> l) & 1) ? gp.at(l) : 0xFFFFFFFFu; }, &c);
        const uint64_t blockers = W::ballot([&](int l) { return ((oddj >> l) & 1) != 0 && gp.at(l) < pmin; });
        if (blockers != 0 || pmin >= gB.bcast(8 * j) || c < 0) break;
        L = fast_uniform(c);
      }
      const uint32_t x = gx.bcast(L);
      const int a = (int)gp.bcast(L);
      const int sj = (int)bslot.bcast(bi);
      if ((g_touched >> L) & 1) {
        // The claim gained pods since the group's test; it is still this entry's first candidate (its place among the others
        // is what it was, or it was chosen by position above): test this one claim again.
        const FastSlot cs = lds_get(&aslot[sj]);
        const FastClaim st = cst.get(x);
        const uint32_t know = okey[a];
        const uint64_t m = st.vmask & cs.cvmask;
        FastEnt e = lds_get(&ent[fast_hash(m)]);
        const int base_ok = (int)((cs.tmplok >> (st.vmask >> 56)) & 1u) & (int)fast_fields_ok(m, cs.dmask);
        int simple = (int)(e.info & 1u) & (int)(e.vmask == m);
        if (base_ok & (simple ^ 1) & (int)(e.info & 1u)) simple = (int)(fast_lookup(ent, m, e) >= 0);   // not the cache's first probe: the probe sequence
        int fit = (int)fast_fits_first(e, st.req, cs.size);
        if (base_ok & simple & (fit ^ 1) & (int)(((e.info >> 8) & 0xFFu) != 0)) fit = (int)fast_fits(pool, e, st.req, cs.size);
        n_tests++;
        if (base_ok & (simple ^ 1)) break;   // the requirement set is not cached yet: the window test raises the event
        if (!(base_ok & simple & fit)) { g_acc &= ~(1ull << L); continue; }   // rejected for good: the entry's next accepting lane
        W::each([&](int l) {
          if (l == L) {
            gk.at(l) = know; gm.at(l) = m;
            gq0.at(l) = (uint32_t)(st.req[0] + cs.size[0]); gq1.at(l) = (uint32_t)(st.req[1] + cs.size[1]);
            gq2.at(l) = (uint32_t)(st.req[2] + cs.size[2]); gq3.at(l) = (uint32_t)(st.req[3] + cs.size[3]);
          }
        });
        g_touched &= ~(1ull << L);
      }
      const uint32_t cnt = gk.bcast(L);
      if (KS_UNLIKELY(cnt >= 65534u)) break;   // the window test raises the event
      const uint32_t mvn = cnt + 1, ua = (uint32_t)a;
      // the claims behind it, for the move of the next add's sort.Slice (scheduler.go:598)
      LaneVar<uint32_t> kv2, ov2;
      const int nm1 = n - 1;
      const uint64_t lessm = W::ballot([&](int l) {
        const int i0 = a + 1 + l, i = i0 < nm1 ? i0 : nm1;   // clamped: no lane is switched off for the two reads
        const uint32_t k = okey[i];
        kv2.at(l) = k; ov2.at(l) = oord[i];
        return (int)(i0 <= nm1) & (int)(k < mvn);
      });
      const int s_ = lessm == ~0ull ? 64 : ctz64(~lessm);   // sorted beyond a: the smaller counts are a prefix
      if (KS_UNLIKELY(s_ >= 64)) break;                      // a long run: the pending path
      // only the move that is one short shift, or none (otherwise: pdqsort's other paths, behind the window test)
      if (KS_UNLIKELY(s_ != 0 && ((int)mid_n | (int)((uint32_t)a - e1 <= 2u) | (int)((uint32_t)a - e2 <= 2u) | (int)((uint32_t)a - e3 <= 2u)))) break;
      // ---- NodeClaim.Add (nodeclaim.go:247-263) on the claim of lane L; the claim lands behind the s_ claims it passes ----
      W::each([&](int l) {
        if (l == L) {
          FastClaim ns;
          ns.vmask = gm.at(l); ns.req[0] = (int32_t)gq0.at(l); ns.req[1] = (int32_t)gq1.at(l); ns.req[2] = (int32_t)gq2.at(l); ns.req[3] = (int32_t)gq3.at(l);
          cst.put(gx.at(l), ns);
        }
        if (l <= s_) { okey[a + l] = (uint16_t)(l == s_ ? mvn : kv2.at(l)); oord[a + l] = (uint16_t)(l == s_ ? x : ov2.at(l)); }
        const bool me = l == bi;
        oclaim.at(l) = me ? x : oclaim.at(l); ocnt.at(l) = me ? cnt : ocnt.at(l);
      });
      n_ref += (unsigned long long)a + 1;
      const uint64_t same = W::ballot([&](int l) { return gx.at(l) == x; });
      g_touched |= same;
      {
        // cursors and the group's positions in (a, a+s_] step left; the class's own cursor comes to a (the claims between it
        // and this one rejected the class for good). Plain arithmetic: (r - a - 1) < s_ as unsigned is a < r <= a + s_.
        const uint32_t ua1 = ua + 1u, su = (uint32_t)s_, usj = (uint32_t)sj;
        W::each([&](int l) {
#pragma unroll
          for (int jj = 0; jj < kFastRows; ++jj) {
            const uint32_t rr = cur[jj].at(l);
            const uint32_t sh = rr - (uint32_t)((rr - ua1) < su);
            cur[jj].at(l) = (uint32_t)(jj * 64 + l) == usj ? ua : sh;
          }
        });
        if (s_) {
          g_jumped |= same;
          const uint32_t b = ua + su;
          W::each([&](int l) {
            const uint32_t pp = gp.at(l), bb = gB.at(l);
            gp.at(l) = ((same >> l) & 1) ? b : pp - (uint32_t)((pp - ua1) < su);
            gB.at(l) = bb - (uint32_t)((bb - ua1) < su);
          });
        }
      }
      W::sync();
      gj++; bi++; steps++;
      placed_any = true;
    }
    KS_SEC(ts2)   // entries placed from the group
    if (gn > 0) {
      if (gj >= gn) { gj = 0; gn = 0; continue; }   // the group is used up (its last entry was placed above)
      gj++;                                           // the window test places this entry; the rest of the group stays valid behind it
    }
    const int slot_w = placed_any ? (int)bslot.bcast(bi) : slot;
    const FastSlot cs = lds_get(&aslot[slot_w]);
    uint32_t rc0 = 0;
    {
      const uint32_t c0 = cur[0].bcast(slot_w & 63), c1 = cur[1].bcast(slot_w & 63), c2 = cur[2].bcast(slot_w & 63), c3 = cur[3].bcast(slot_w & 63);
      const int row = slot_w >> 6;
      rc0 = row == 0 ? c0 : row == 1 ? c1 : row == 2 ? c2 : c3;
    }
    static_assert(kFastRows == 4, "cursor rows are spelled out above");
    uint32_t r = rc0;
    int outcome = 0;   // 1 placed, 2 no acceptor, 3 event
    while ((int)r < n) {
      // ---- addToInflightNode (scheduler.go:658-692): positions r .. r+63, one lane each; straight-line: three
      // dependent LDS reads (order -> claim state -> requirement-set cache), everything else in registers ----
      LaneVar<uint64_t> mv;
      LaneVar<uint32_t> xv, kv;
      LaneVar<int32_t> q0, q1, q2, q3;
      const uint32_t r0 = r;
      uint64_t okm = 0, oddm = 0;
      W::ballot2([&](int l) {
        const int p = (int)r0 + l;
        const bool valid = p < n;
        const int pc = valid ? p : n - 1;
        const uint32_t x = oord[pc];
        xv.at(l) = x; kv.at(l) = valid ? (uint32_t)okey[pc] : 0xFFFFFFFFu;
        const FastClaim st = cst.get(x);
        q0.at(l) = st.req[0]; q1.at(l) = st.req[1]; q2.at(l) = st.req[2]; q3.at(l) = st.req[3];
        const uint64_t m = st.vmask & cs.cvmask;
        mv.at(l) = m;
        const FastEnt e = lds_get(&ent[fast_hash(m)]);
        // predicates as 0/1 integers combined with & and |: straight-line code, no short-circuit branches
        const int base_ok = (int)valid & (int)((cs.tmplok >> (st.vmask >> 56)) & 1u) & (int)fast_fields_ok(m, cs.dmask);
        const int simple = (int)(e.info & 1u) & (int)(e.vmask == m);   // the cache's first probe is this requirement set
        const int fit = (int)fast_fits_first(e, st.req, cs.size);
        // bit 0: accepted; bit 1: needs the long way (requirement set not cached, a hash collision, or further Pareto vectors)
        return (base_ok & simple & fit) | ((base_ok & ((simple ^ 1) | ((fit ^ 1) & (int)(((e.info >> 8) & 0xFFu) != 0)))) << 1);
      }, okm, oddm);
      uint64_t missm = 0;
      oddm &= okm ? (okm & (0ull - okm)) - 1ull : ~0ull;   // only the lanes before the first plain acceptor can change the answer
      if (KS_UNLIKELY(oddm != 0)) {
        // rare: resolve those lanes with the full probe sequence / all Pareto vectors
        const uint64_t mm = oddm;
        uint64_t ok2 = 0;
        W::ballot2([&](int l) {
          if (!((mm >> l) & 1)) return 0;
          FastEnt e;
          if (fast_lookup(ent, mv.at(l), e) < 0) return 2;
          const int32_t rq[4] = {q0.at(l), q1.at(l), q2.at(l), q3.at(l)};
          return fast_fits(pool, e, rq, cs.size) ? 1 : 0;
        }, ok2, missm);
        okm |= ok2;
      }
      n_tests += (unsigned long long)(n - (int)r0 < 64 ? n - (int)r0 : 64);
      n_steps++;
      const int first_ok = okm ? ctz64(okm) : 64;
      if (KS_UNLIKELY(missm != 0 && ctz64(missm) < first_ok)) {
        // a requirement set that is not cached yet sits before the first acceptor: cache it, test these positions again
        ev = FEV_ENTRY; ev_vm = mv.bcast(ctz64(missm)); outcome = 3;
        break;
      }
      if (!okm) { r = (uint32_t)((int)r0 + 64 < n ? (int)r0 + 64 : n); continue; }
      // ---- commit: NodeClaim.Add (nodeclaim.go:247-263) ----
      const int a = (int)r0 + first_ok;
      const int x = (int)xv.bcast(first_ok);
      const uint32_t cnt = kv.bcast(first_ok);
      if (KS_UNLIKELY(cnt >= 65534u)) { ev = FEV_COUNT; outcome = 3; break; }
      FastClaim ns;
      ns.vmask = mv.bcast(first_ok);
      ns.req[0] = q0.bcast(first_ok) + cs.size[0]; ns.req[1] = q1.bcast(first_ok) + cs.size[1];
      ns.req[2] = q2.bcast(first_ok) + cs.size[2]; ns.req[3] = q3.bcast(first_ok) + cs.size[3];
      if (W::leader()) cst.put(x, ns);
      oclaim.set(bi, (uint32_t)x); ocnt.set(bi, cnt);
      n_ref += (unsigned long long)a + 1;
      r = (uint32_t)a;
      outcome = 1;
      // The sort.Slice of the NEXT add (scheduler.go:598) repairs this claim's position: one stable move past the claims
      // with a smaller count. When the next add follows inside this block and those claims are all among the positions
      // just tested, their counts and ids are in registers already: move now, without reading the order again.
      const uint32_t mvn = cnt + 1;
      bool moved = false;
      if (KS_LIKELY(bi + 1 < bn && !(max_steps >= 0 && steps + 1 >= max_steps))) {
        const uint64_t lessm = W::ballot([&](int l) { return l > first_ok && kv.at(l) < mvn; });   // lanes past n hold 0xFFFFFFFF
        const uint64_t t = first_ok == 63 ? 0ull : (lessm >> (first_ok + 1));
        const int s_ = t == ~0ull ? 64 : ctz64(~t);
        // the single stable move — or no move at all (the next claim has at least the new count: sorted as it stands)
        if (KS_LIKELY((first_ok + 1 + s_ < 64 || (int)r0 + 64 >= n) && (s_ == 0 || n <= 12 || (n >= 50 && !fast_sampled(n, a))))) {
          // lanes first_ok+1 .. first_ok+s_ step one position to the left, the claim lands behind them
          W::each([&](int l) { if (l > first_ok && l <= first_ok + s_) { okey[(int)r0 + l - 1] = (uint16_t)kv.at(l); oord[(int)r0 + l - 1] = (uint16_t)xv.at(l); } });
          if (W::leader()) { okey[a + s_] = (uint16_t)mvn; oord[a + s_] = (uint16_t)x; }
          const int b = a + s_;
          W::each([&](int l) {
#pragma unroll
            for (int j = 0; j < kFastRows; ++j) { const uint32_t rr = cur[j].at(l); cur[j].at(l) = rr - (uint32_t)(((uint32_t)a < rr && rr <= (uint32_t)b) ? 1 : 0); }
          });
          if (gj < gn) {   // the group's lanes follow the move; the claim's acceptances there are void
            const uint64_t same = W::ballot([&](int l) { return gx.at(l) == (uint32_t)x; });
            g_touched |= same;
            if (s_) g_jumped |= same;
            W::each([&](int l) {
              const uint32_t pp = gp.at(l), bb = gB.at(l);
              gp.at(l) = ((same >> l) & 1) ? (uint32_t)b : pp - (uint32_t)(((uint32_t)a < pp && pp <= (uint32_t)b) ? 1 : 0);
              gB.at(l) = bb - (uint32_t)(((uint32_t)a < bb && bb <= (uint32_t)b) ? 1 : 0);
            });
          }
          moved = true;
        }
      }
      if (KS_UNLIKELY(!moved)) {
        if (W::leader()) okey[a] = (uint16_t)mvn;
        pend_a = a; pend_x = x; pend_mv = mvn;
        gj = 0; gn = 0;
      }
      W::sync();
      break;
    }
    if (KS_UNLIKELY(outcome == 3)) {
      // the event interrupts this pod: what the scan learned (claims that rejected it for good) is kept in its cursor
      W::each([&](int l) {
#pragma unroll
        for (int j = 0; j < kFastRows; ++j) if (j * 64 + l == slot_w) cur[j].at(l) = r;
      });
      break;
    }
    if (r != rc0) W::each([&](int l) {
#pragma unroll
      for (int j = 0; j < kFastRows; ++j) if (j * 64 + l == slot_w) cur[j].at(l) = r;
    });
    steps++;
    KS_SEC(ts3)   // entry placed by the window test
#ifdef KSOLVE_PHASE_TIMERS
    ts7++;
#endif
    if (KS_LIKELY(outcome == 1)) { bi++; continue; }
    ev = FEV_NEWCLAIM; ev_arg = slot_w;   // no in-flight claim accepted the pod: addToNewNodeClaim; the driver moves on to the next pod
    break;
  }
  if (ev == FEV_DONE && bn > 0) {
    // the deadline / a cancellation stopped the loop inside a block: the pods placed so far are results too
    const int dn = bi < bn ? bi : bn, b0 = base;
    W::each([&](int l) { if (l < dn) { gqclaim[b0 + l] = oclaim.at(l); gqcnt[b0 + l] = ocnt.at(l); } });
    bn = 0; bi = 0;
  }
  // ---- state out ----
  if (W::leader()) {
    hs->base = base; hs->bi = bi; hs->bn = bn; hs->n = n; hs->steps = steps; hs->status = status;
    hs->pend_a = pend_a; hs->pend_x = pend_x; hs->pend_mv = pend_mv; hs->pend_new = pend_new ? 1 : 0;
    hs->n_steps = n_steps; hs->n_tests = n_tests; hs->n_ref = n_ref; hs->ev_arg = ev_arg; hs->ev_vm = ev_vm;
    hs->hot_cycles += W::clock() - t_in;
#ifdef KSOLVE_PHASE_TIMERS
    hs->tsec[0] += ts0; hs->tsec[1] += ts1; hs->tsec[2] += ts2; hs->tsec[3] += ts3; hs->tsec[4] += ts4; hs->tsec[5] += ts5; hs->tsec[6] += ts6; hs->tsec[7] += ts7;
#endif
  }
#undef KS_SEC
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < kFastRows; ++j) hs->cur[j][l] = cur[j].at(l);
    hs->nxt_cls[l] = nxt_cls.at(l); hs->bcls[l] = bcls.at(l);
    hs->bslot[l] = bslot.at(l); hs->oclaim[l] = oclaim.at(l); hs->ocnt[l] = ocnt.at(l);
  });
  W::sync();
  return ev;
}

// The driver: runs the loop, handles its events through FastCold.
template <class W, int GS = 0>
struct FastEngine {
  FastCold<W, GS> cold;
  KS_LDS FastHot* hs;
  KS_DEV FastEngine(const ProblemView* p, const Workspace* s, const FastWork* f, char* lds) { cold.init(p, s, f, lds); hs = (KS_LDS FastHot*)(lds + f->plan.off_hot); }

  KS_DEV void solve() {
    {
      const int why = (int)W::uniform((uint64_t)(uint32_t)cold.setup());
      if (why) { cold.bail_code = why; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
    }
    KS_LDS FastHot* const h = fast_uniform(hs);
    const int np = fast_uniform(cold.Pk->n_pods);
    if (W::leader()) {
      h->base = 0; h->bi = 0; h->bn = 0; h->n = 0; h->np = np; h->steps = 0; h->status = 0;
      const long long ms_ = cold.Sk->max_steps;
      h->max_steps = ms_ < 0 ? -1 : (int)(ms_ > 0x7FFFFFFF ? 0x7FFFFFFF : ms_);
      h->pend_a = -1; h->pend_x = 0; h->pend_mv = 0; h->pend_new = 0; h->ev_arg = 0; h->ev_vm = 0;
      h->n_steps = 0; h->n_tests = 0; h->n_ref = 0; h->hot_cycles = 0;
      for (int i = 0; i < 8; ++i) h->tsec[i] = 0;
      h->q_class = cold.Fk->q_class; h->cancel = cold.Sk->cancel_flag;
      h->q_claim = cold.Fk->q_claim; h->q_cnt = cold.Fk->q_cnt;
    }
    {
      const uint32_t* qc = cold.Fk->q_class;
      W::each([&](int l) {
        for (int j = 0; j < kFastRows; ++j) h->cur[j][l] = 0;
        h->nxt_cls[l] = l < np ? qc[l] : 0; h->bcls[l] = 0; h->bslot[l] = 0xFFFFu; h->oclaim[l] = 0; h->ocnt[l] = 0;
      });
    }
    W::sync();
    FastHotCtx<GS> cx;
    cx.okey = cold.order.key; cx.oord = cold.order.ord; cx.cst = cold.cst; cx.ent = cold.ent; cx.pool = cold.pool;
    cx.aslot = cold.aslot; cx.slot_of = cold.Mp->slot_of; cx.hs = hs;
    unsigned long long tev[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = W::clock();
    for (;;) {
      const int ev = fast_uniform(fast_hot_run<W, GS>(cx));
      if (ev == FEV_DONE) break;
      const unsigned long long te0 = W::clock();
      if (ev == FEV_ENTRY) {
        if (fast_uniform(cold.create_entry(h->ev_vm)) < 0) { cold.bail_code = 21; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
      } else if (ev == FEV_SLOT) {
        const int k = fast_uniform(h->ev_arg);
        const int sv = fast_uniform(cold.new_slot(k));
        const int slot = sv & 0xFFFF;
        W::each([&](int l) {
          if (sv >> 16) {   // every slot was taken: all classes start over
            h->bslot[l] = 0xFFFFu;
            for (int j = 0; j < kFastRows; ++j) h->cur[j][l] = 0;
          }
          if (h->bcls[l] == (uint32_t)k && l < h->bn) h->bslot[l] = (uint32_t)slot;   // this pod and later pods of the class in the block
          if (l == (slot & 63)) h->cur[slot >> 6][l] = 0;
        });
        W::sync();
      } else if (ev == FEV_SLOWSORT || ev == FEV_PLACE) {
        const int n = fast_uniform(h->n);
        int b = -1;
        if (ev == FEV_SLOWSORT) cold.slow_sort(n, fast_uniform(h->ev_arg), 0);
        else b = fast_uniform(cold.place_new_claim(n));
        if (b == -2) { cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
        if (b == -1) {
          // pdqsort permuted positions lo..hi: cursors inside fall back to lo
          const int lo = fast_uniform(cold.lo_), hi = fast_uniform(cold.hi_);
          if (hi >= lo) W::each([&](int l) { for (int j = 0; j < kFastRows; ++j) { const uint32_t r = h->cur[j][l]; if (r > (uint32_t)lo && r <= (uint32_t)hi) h->cur[j][l] = (uint32_t)lo; } });
        } else {
          // positions [b, n-1) moved right by one; a cursor past b either steps over the new claim or — if its class is
          // accepted by it — comes back to it
          const KS_LDS uint64_t* acc = cold.Mp->acc;
          W::each([&](int l) { for (int j = 0; j < kFastRows; ++j) { const uint32_t r = h->cur[j][l]; if (r > (uint32_t)b) h->cur[j][l] = ((acc[j] >> l) & 1) ? (uint32_t)b : r + 1; } });
        }
        if (W::leader()) { h->pend_a = -1; h->pend_new = 0; }
        W::sync();
      } else if (ev == FEV_NEWCLAIM) {
        const int n = fast_uniform(h->n), bi = fast_uniform(h->bi);
        const int made = fast_uniform(cold.new_claim(fast_uniform(h->ev_arg), bi, n));
        if (!made) { cold.finish(fast_uniform(cold.bail_code) < 0 ? 1 : 3, 0, (unsigned long long)fast_uniform(h->steps), 0, 0, 0, nullptr); return; }
        if (W::leader()) { h->oclaim[bi] = (uint32_t)n; h->ocnt[bi] = 0; h->n = n + 1; h->pend_new = 1; h->bi = bi + 1; }   // claim ids are handed out in creation order
        W::sync();
      } else { cold.bail_code = 22; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
      if (ev >= 1 && ev <= 5) { tev[ev] += W::clock() - te0; nev[ev]++; }
    }
    // profiling builds (-DKSOLVE_PHASE_TIMERS): cycles inside the loop function, per event kind, in total; event counts
    unsigned long long tc[16] = {h->hot_cycles, tev[1], tev[2], tev[3], tev[4], tev[5], W::clock() - t_begin, nev[3] + (nev[4] << 20) + (nev[1] << 40),
                                 h->tsec[0], h->tsec[1], h->tsec[2], h->tsec[3], h->tsec[4], h->tsec[5], h->tsec[6], h->tsec[7]};
    cold.finish(fast_uniform(h->status), fast_uniform(h->n), (unsigned long long)fast_uniform(h->steps), h->n_steps, h->n_tests, h->n_ref, tc);
  }
};

// ksolve_fast_queue — one thread per queue entry, before the loop: its class, and "not placed"
// ksolve_fast_scatter — one thread per queue entry, after the loop: the entry's result under its pod index (Results.pod_assignment / pod_slot)
struct FastQueueArgs {
  const uint32_t* sorted; const uint32_t* row_class;
  uint32_t* q_class; uint32_t* q_claim; uint32_t* q_cnt;
  int32_t* assign; uint32_t* slot;
};
KS_FN void fast_queue_body(int i, const FastQueueArgs& a) { a.q_class[i] = a.row_class[a.sorted[i]]; a.q_claim[i] = 0xFFFFFFFFu; }
KS_FN void fast_scatter_body(int i, const FastQueueArgs& a) {
  const uint32_t c = a.q_claim[i];
  if (c == 0xFFFFFFFFu) return;
  const uint32_t p = a.sorted[i];
  a.assign[p] = (int32_t)c; a.slot[p] = a.q_cnt[i];
}

// ksolve_fast_records — one wavefront per claim: materialises the hot claim record the finalize kernel and the result
// download read (ksp.h RecLayout) from the cursor engine's compact state: requirement masks = the template's with the
// variable keys' fields, InstanceTypeOptions = F(requirement set) ∩ { allocatable >= requests }.
struct FastRecordArgs {
  ProblemView pv;
  Workspace ws;
  FastWork fw;
};
template <class W>
KS_DEV void fast_record_body(int c, const FastRecordArgs& a) {
  const ProblemView& P = a.pv;
  const RecLayout ly = P.lay;
  const Dict& d = P.dict;
  const FastClaim st = a.fw.c_state[c];
  const int t = (int)(st.vmask >> 56);
  const FastVar fv = *a.fw.var;
  // keys the requirement set defines: the template's, and of the keys pods select on those whose guard bit is clear
  uint32_t vdef = P.tmpl_reqs.defined[t];
  for (int j = 0; j < fv.nv; ++j) {
    const uint32_t kb = 1u << fv.vkey[j];
    vdef = ((st.vmask >> (fv.voff[j] + fv.vwidth[j])) & 1) ? (vdef & ~kb) : (vdef | kb);
  }
  uint64_t* rec = a.ws.c_hot + (size_t)c * ly.c_hot_words();
  const uint64_t* tm = P.tmpl_reqs.mask + (size_t)t * d.req_words;
  W::for_n(ly.rw, [&](int w) {
    uint64_t v = tm[w];
    for (int j = 0; j < fv.nv; ++j) if (fv.vword[j] == w && ((vdef >> fv.vkey[j]) & 1u)) v = (st.vmask >> fv.voff[j]) & ((1ull << fv.vwidth[j]) - 1);
    rec[ly.c_mask() + w] = v;
  });
  const int iw = ly.iw, nr = ly.nr, n_its = P.n_its;
  const uint64_t* eits = a.fw.ent_its + (size_t)a.fw.c_ent[c] * iw;
  for (int j = 0; j < iw; ++j) {
    const uint64_t in = eits[j];
    const uint64_t okm = in ? W::ballot([&](int l) {
      const int it = j * 64 + l;
      if (it >= n_its || !((in >> l) & 1)) return false;
      bool f = true;
      for (int r = 0; r < nr; ++r) f = f && (int64_t)st.req[r] <= P.it_alloc[(size_t)r * n_its + it];
      return f;
    }) : 0ull;
    W::store(&rec[ly.c_its() + j], okm);
  }
  W::for_n(nr, [&](int r) { rec[ly.c_total() + r] = (uint64_t)(int64_t)st.req[r]; rec[ly.c_head() + r] = 0; });
  if (W::leader()) {
    rec[ly.c_f0()] = (uint64_t)vdef;
    rec[ly.c_f1()] = 0;
    rec[ly.c_meta()] = (uint64_t)(uint32_t)t | ((uint64_t)a.fw.c_npods[c] << 32);
    rec[ly.c_meta2()] = (uint64_t)a.fw.c_hostseq[c];
  }
  W::sync();
}

}  // namespace ks
