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

constexpr int kFastRows = 4;          // class slots = 64 lanes x R rows, R = 1 or kFastRows (FastPlan::rows; the engine is compiled for both)
constexpr uint32_t kFastFree = 0xFFFFFFFFu;   // class id of a free slot
constexpr uint32_t kFastLastBit = 0x80000000u;   // q_class: the class's last entry in the queue (its slot is free afterwards)
constexpr int kFastEnt = 1024;        // requirement-set cache slots (open addressing, filled to 80% at most)
constexpr int kFastPool = 256;        // extra Pareto vectors
constexpr int kFastMaxPareto = 16;    // per requirement set
constexpr int kFastMaxVar = 12;       // keys pods select on
constexpr uint64_t kFastExtBit = 1ull << 63;   // FastEnt::vmask of an entry with further Pareto vectors (template ids are < 32: the bit is free): "the slot's key equals the set"
                                                // is then false in the loops' one-compare test, and the entry is read the long way (fast_lookup + fast_fits)
constexpr int kFastVarBits = 56;      // their dictionary values + one guard bit each must fit 56 bits; the top byte of vmask is the template

struct FastClaim { uint64_t vmask; int32_t req[4]; };                                   // 24 B: a claim's state (requirement set, requests)
// The in-flight claim as the loop keeps it, by claim id: its state and its ACCEPTANCE WORDS — bit s of acc[j] says that the
// class in slot 64 j + s passes CanAdd on this claim as it stands (nodeclaim.go:124-242). The words are exact at all times:
// they are recomputed for the one claim a commit changes (64 classes per instruction, one lane each), for a new claim when it
// is created, and for one class over all claims when the class takes a slot.
template <int R> struct FastRec { uint64_t vmask; int32_t req[4]; uint64_t acc[R]; };  // 32 B (R = 1) / 56 B (R = 4)
// a pod class as the scan needs it: values it admits (all ones on keys it does not select on), the fields it selects on,
// requests, templates whose taints it tolerates and whose keys cover its custom keys, keys it defines
struct FastSlot { uint64_t cvmask; uint64_t dmask; int32_t size[4]; uint32_t tmplok; uint32_t kdef; };  // 40 B
struct FastEnt { uint64_t vmask; int32_t cap[4]; uint32_t info; uint32_t pad; };        // 32 B: info bit0 valid, bits 8..15 extra vectors, bits 16..31 pool offset

struct FastPlan {   // LDS plan of ksolve_pack_fast (bytes), computed by the host
  int total_bytes, cap;
  int off_state, off_key, off_ord, off_snap, off_ent, off_pool, off_slot, off_misc, off_hot;
  int global_state;   // 1: the claims' records (FastRec) live in HBM (FastWork::c_rec) and so does the snapshot a slow sort compares
                      //    with (FastWork::o_snap); only the order's keys and ids (4 B per claim) in LDS: ~27,000 in-flight claims
                      //    instead of ~3,000 (round 4: ~15,000 with the snapshot in LDS too). off_state / off_snap are unused then.
                      // 2: the order arrays too (FastWork::o_key / o_ord / o_snap): 65,472 claims — what 16-bit claim ids address;
                      //    LDS holds the caches, the class slots and the loop's own state only.
  int rows;           // class slots / 64: 1 when at most 64 pod classes are ever live at once in the queue (FastWork::max_active,
                      //    counted before the loop by ksolve_fast_overlap), kFastRows otherwise
  int helper;         // 1: the two-wavefront kernel (ksolve_pack_fast2: plan 0, one row of class slots) — a second wavefront of the
                      //    workgroup recomputes a claim's acceptance words while the first one goes on to the next pod (FastMail)
};

struct FastMisc {   // small LDS tables
  uint64_t tvmask[32];                 // template: values it admits on the variable keys | template << 56
  uint32_t tdef[32];
  uint64_t fmask[kFastMaxVar];         // field of variable key j inside vmask
  uint8_t vkey[kFastMaxVar], voff[kFastMaxVar], vwidth[kFastMaxVar];
  uint16_t vword[kFastMaxVar];         // dictionary word of the key
  uint64_t its[kMaxItWords], rem[kMaxItWords], cand[kMaxItWords];   // slow-path scratch
  uint64_t acc[kFastRows];             // acceptance words of the claim just created (new_claim -> place_new_claim)
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
  uint32_t* q_class;      // [n_pods] class of queue entry i = row_class[sorted_pods[i]] | kFastLastBit on the class's last entry
  uint32_t* q_claim;      // [n_pods] claim of queue entry i (0xFFFFFFFF: not placed)
  uint32_t* q_cnt;        // [n_pods] pods the claim held before it
  uint16_t* o_key;        // [max_claims] plan 2: the claim order (pod count / claim id by position) and its snapshot, in HBM
  uint16_t* o_ord;
  uint16_t* o_snap;
  uint64_t* c_rec;        // plans 1, 2: [max_claims] FastRec<rows>, the claims' records in HBM
  uint32_t* cls_first;    // [n_classes] first / last queue entry of the class (ksolve_fast_queue)
  uint32_t* cls_last;
  uint32_t* max_active;   // [1] the most classes live at once: max over classes c of #{c' : first(c') <= first(c) <= last(c')} (ksolve_fast_overlap)
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


// ... and as 16-byte pieces for records that are multiples of 16 bytes at 16-byte-aligned addresses (FastEnt, FastRec<1>; the plan
// aligns every table to 16): ds_read_b128 takes 4 LDS cycles per wave-instruction where ds_read2_b64 takes 8, ds_write_b128 13 where
// two ds_write_b64 take 12 + 12 (MI355X_MICROARCH.md, LDS table).
typedef uint32_t __attribute__((vector_size(16), may_alias)) u32x4_alias;
template <class T>
KS_FN T lds_get16(const KS_LDS T* p) {
  static_assert(sizeof(T) % 16 == 0, "lds_get16: 16-byte multiples");
  T out;
  u32x4_alias* o = (u32x4_alias*)&out;
  const KS_LDS u32x4_alias* s = (const KS_LDS u32x4_alias*)p;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 16); ++i) o[i] = s[i];
  return out;
}
// the live 24 bytes of a cache entry (key, capacities): one 16-byte and one 8-byte read — no destination register is dead, so the
// register allocator cannot hand one to the next instruction while the read is in flight (a write-after-write wait)
KS_FN FastEnt ent_live(const KS_LDS FastEnt* p) {
  FastEnt out;
  *(u32x4_alias*)&out = *(const KS_LDS u32x4_alias*)p;
  *(u64_alias*)&out.cap[2] = *(const KS_LDS u64_alias*)&p->cap[2];
  out.info = 1; out.pad = 0;
  return out;
}
template <class T>
KS_FN void lds_put16(KS_LDS T* p, const T& v) {
  static_assert(sizeof(T) % 16 == 0, "lds_put16: 16-byte multiples");
  const u32x4_alias* o = (const u32x4_alias*)&v;
  KS_LDS u32x4_alias* s = (KS_LDS u32x4_alias*)p;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 16); ++i) s[i] = o[i];
}

// The in-flight claims' records by claim id. LDS while the problem's claims fit beside the order arrays and the caches (the
// benchmarked configuration: 2,763 claims); HBM otherwise (HBM = true): the lane that reads a claim then gathers its record
// through the vector L1 / L2 instead of LDS. Plain loads and stores: this wavefront is the only reader and writer, its vector
// memory operations execute in order, W::sync() (a wavefront-scope fence) follows every store — what the general engine's
// HBM-resident claim records have relied on since round 1.
template <bool HBM, int R> struct ClaimRecs;
template <int R> struct ClaimRecs<false, R> {
  KS_LDS FastRec<R>* p;
  KS_FN FastClaim state(uint32_t c) const { return lds_get((const KS_LDS FastClaim*)&p[c]); }
  KS_FN void put_state(uint32_t c, const FastClaim& v) const { lds_put((KS_LDS FastClaim*)&p[c], v); }
  KS_FN uint64_t acc(uint32_t c, int row) const { return p[c].acc[row]; }
  KS_FN void put_acc(uint32_t c, int row, uint64_t v) const { p[c].acc[row] = v; }
  // the whole record in one go (the select step of the fast loop reads state and acceptance words together)
  KS_FN FastRec<R> rec(uint32_t c) const { if constexpr (sizeof(FastRec<R>) % 16 == 0) return lds_get16(&p[c]); else return lds_get(&p[c]); }
  KS_FN void put_rec(uint32_t c, const FastRec<R>& v) const { if constexpr (sizeof(FastRec<R>) % 16 == 0) lds_put16(&p[c], v); else lds_put(&p[c], v); }
};
template <int R> struct ClaimRecs<true, R> {
  FastRec<R>* p;
  KS_FN FastClaim state(uint32_t c) const {
    FastClaim out;
    u64_alias* o = (u64_alias*)&out;
    const u64_alias* s = (const u64_alias*)&p[c];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FastClaim) / 8); ++i) o[i] = s[i];
    return out;
  }
  KS_FN void put_state(uint32_t c, const FastClaim& v) const {
    const u64_alias* o = (const u64_alias*)&v;
    u64_alias* s = (u64_alias*)&p[c];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(FastClaim) / 8); ++i) s[i] = o[i];
  }
  KS_FN uint64_t acc(uint32_t c, int row) const { return p[c].acc[row]; }
  KS_FN void put_acc(uint32_t c, int row, uint64_t v) const { p[c].acc[row] = v; }
  KS_FN FastRec<R> rec(uint32_t c) const { return p[c]; }
  KS_FN void put_rec(uint32_t c, const FastRec<R>& v) const { p[c] = v; }
};
// The plan the engine is compiled for (FastPlan::global_state): 0 everything in LDS, 1 claim records in HBM, 2 claim records and
// order arrays in HBM. The order's accesses are plain loads and stores: the wavefront is the only reader and writer, its vector
// memory operations execute in order, and W::sync() (a wavefront-scope fence) stands between a store and another lane's load.
template <int GS, int R> struct FastMem {
  static constexpr bool kStateHbm = GS >= 1, kOrderHbm = GS >= 2, kSnapHbm = GS >= 1;   // (the snapshot around a slow sort is cold: in HBM as soon as LDS is short)
#if KS_DEVICE
  typedef typename std::conditional<kOrderHbm, uint16_t*, KS_LDS uint16_t*>::type o16;
#else
  typedef uint16_t* o16;
#endif
#if KS_DEVICE
  typedef typename std::conditional<kSnapHbm, uint16_t*, KS_LDS uint16_t*>::type s16;
#else
  typedef uint16_t* s16;
#endif
  typedef ClaimRecs<kStateHbm, R> States;
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

template <bool H, int R> KS_FN ClaimRecs<H, R> fast_uniform(ClaimRecs<H, R> c) { c.p = fast_uniform(c.p); return c; }

#define KS_LIKELY(x) __builtin_expect(!!(x), 1)
#define KS_UNLIKELY(x) __builtin_expect(!!(x), 0)
#if KS_DEVICE
#define KS_COLD __device__ __attribute__((noinline))
#else
#define KS_COLD inline
#endif

// requirement-set cache helpers (per lane)
// One 32-bit multiply (a quarter-rate instruction: 16 cycles of a lone wavefront; the 64-bit product of the first version was three of
// them): the two halves folded with a rotation — the template id sits in the top byte, the fields of the keys pods select on
// fill the word from bit 0 — and the TOP ten bits of the product, the only ones every input bit reaches.
KS_FN uint32_t fast_hash(uint64_t vm) {
  const uint32_t lo = (uint32_t)vm, hi = (uint32_t)(vm >> 32);
  return (((lo ^ ((hi << 15) | (hi >> 17))) * 0x9E3779B1u) >> 22) & (kFastEnt - 1);
}
// entry of requirement set vm (copied to `out`), or -1 (not cached yet); one 32-byte LDS read per probe
KS_FN int fast_lookup(const KS_LDS FastEnt* ent, uint64_t vm, FastEnt& out) {
  uint32_t h = fast_hash(vm);
  for (int probe = 0; probe < kFastEnt; ++probe) {
    out = lds_get(&ent[h]);
    if (!(out.info & 1u)) return -1;
    if ((out.vmask & ~kFastExtBit) == vm) return (int)h;
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
// the smaller of two wave-uniform values, in scalar registers (left to itself the compiler builds v_min3_u32 out of three of these:
// two moves into vector registers, the minimum, a readfirstlane back)
KS_FN uint32_t fast_umin_s(uint32_t a, uint32_t b) {
#if KS_DEVICE
  uint32_t r;
  asm("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
  return r;
#else
  return a < b ? a : b;
#endif
}
KS_FN bool fast_sampled(int n, int p) {   // choosePivot's nine positions (pdq_emul.h sort()); n >= 50
  const unsigned q = (unsigned)n >> 2, u = (unsigned)p;
  return u - (q - 1) <= 2u || u - (2 * q - 1) <= 2u || u - (3 * q - 1) <= 2u;
}

// Everything that happens rarely (a new requirement set, a new claim, a new class slot, pdqsort leaving its single-move
// path): real function calls, so that their code and registers stay out of the loop that places a pod.
// The loop's state between two events (LDS): scalars, the class slots (lane = slot: class id, cursor) and the 64-entry queue block
// The mailbox between the wavefront that places pods ("placer") and the one that recomputes acceptance words ("refresher") in the
// two-wavefront kernel (ksolve_pack_fast2). A ring of two requests: request s (1, 2, ...) is ONE word, (s << 16) | claim, in
// word[s & 1], written after the claim's new state (NodeClaim.Add) went to its record; the refresher serves the requests in order:
// the claim's state from its record, CanAdd of that claim for the classes of all slots, the claim's acceptance word (or, when a
// requirement set is not cached, `miss_x` of the ring slot and the sticky low bit of `done`: the word stays as it was, the placer
// leaves its loop and the driver computes it), then `done` = s << 1 | miss.
// Between a request and its `done` the claim's word is STALE, and a stale word is a superset of the true one (a claim that gained a
// pod accepts no class it rejected before: requests only grow, the instance-type set only shrinks — the monotonicity the cursors
// rest on): the placer's select may run on it, and waits only when the first acceptor it finds IS a claim with a request in flight.
// The LDS executes every wavefront's accesses in the order they were issued, so whoever sees the second of two words written in
// order sees the first.
struct FastMail {
  uint32_t word[2];   // request s: (s << 16) | claim, in word[s & 1]
  uint32_t done;      // (s << 1) | miss: every request up to s is served
  uint32_t gen;       // bumped by the placer at every entry into its loop (the class slots may have changed; a reported miss is taken)
  uint32_t quit;      // the kernel is over
  int32_t miss_x[2];
  uint32_t helper;    // which wavefront of the workgroup is the refresher (the kernel picks one on another SIMD than the placer's)
  uint32_t simd[4];   // SIMD of each wavefront of the workgroup (HW_ID), written before the barrier
};
#if KS_DEVICE
KS_FN uint32_t mail_load(const KS_LDS uint32_t* p) { return *(const volatile KS_LDS uint32_t*)p; }
KS_FN void mail_store(KS_LDS uint32_t* p, uint32_t v) { *(volatile KS_LDS uint32_t*)p = v; }
#else
KS_FN uint32_t mail_load(const uint32_t* p) { return *p; }
KS_FN void mail_store(uint32_t* p, uint32_t v) { *p = v; }
#endif

struct FastHot {
  alignas(16) FastMail mail;
  int base, bi, bn, n, np, max_steps, steps, status;
  int pend_a, pend_x, pend_new, ev_arg;
  uint32_t pend_mv;
  int rf_x;   // a claim whose acceptance words the driver must compute (the fast loop met a requirement set that is not cached), -1 = none
  int rf_x2;  // two-wavefront kernel: a second one (the claim placed while the refresher reported the first)
  unsigned long long n_steps, n_tests, n_ref, hot_cycles;
  unsigned long long tsec[8];   // profiling builds: shader clock per path of the loop
  unsigned long long hw[4];     // profiling builds, two wavefronts: waits for the claim of a request in flight (count, cycles), for a ring slot (count), cycles from a request to its `done` as seen at those waits
  const uint32_t* q_class; const volatile int* cancel; uint32_t* q_claim; uint32_t* q_cnt;
  uint32_t cur[kFastRows][64];    // cursor of the class in the slot: every claim left of it has rejected the class for good
  uint32_t scls[kFastRows][64];   // class in the slot, kFastFree = none
  uint32_t bcls[64], oclaim[64], ocnt[64], nxt_cls[64];
};

template <class W, int GS, int R>
struct FastCold {
  const ProblemView* Pk;
  const Workspace* Sk;
  const FastWork* Fk;
  typedef typename FastMem<GS, R>::o16 o16;
  ClaimOrder<W, o16, false> order;
  typename FastMem<GS, R>::s16 snap;   // [cap] order snapshot around a slow sort (HBM on plans 1 and 2)
  typename FastMem<GS, R>::States cst;
  KS_LDS FastEnt* ent;
  KS_LDS int32_t* pool;     // [kFastPool][4]
  KS_LDS FastSlot* aslot;   // [64 R] the class of each slot
  KS_LDS FastMisc* Mp;
  KS_LDS FastHot* hs;
  int n_claims = 0, nv = 0, n_ent = 0, n_pool = 0, n_evict = 0;
  uint32_t host_seq = 0, active_templates = 0;
  int bail_code = 0;
  int lo_ = 0, hi_ = -1;    // positions a slow sort permuted
  unsigned long long n_ref_extra = 0, n_cold_tests = 0;

  KS_DEV void init(const ProblemView* p, const Workspace* s, const FastWork* f, char* lds) {
    Pk = p; Sk = s; Fk = f;
    const FastPlan& pl = f->plan;
    Mp = (KS_LDS FastMisc*)(lds + pl.off_misc);
    hs = (KS_LDS FastHot*)(lds + pl.off_hot);
    if constexpr (GS >= 1) cst.p = (FastRec<R>*)f->c_rec; else cst.p = (KS_LDS FastRec<R>*)(lds + pl.off_state);
    typedef typename FastMem<GS, R>::s16 s16;
    if constexpr (GS >= 2) { order.key = (o16)f->o_key; order.ord = (o16)f->o_ord; }
    else { order.key = (o16)(lds + pl.off_key); order.ord = (o16)(lds + pl.off_ord); }
    if constexpr (GS >= 1) snap = (s16)f->o_snap; else snap = (s16)(lds + pl.off_snap);
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
      e.vmask = count > 1 ? (vm | kFastExtBit) : vm; e.cap[0] = f0; e.cap[1] = f1; e.cap[2] = f2; e.cap[3] = f3;
      e.info = 1u | ((uint32_t)(count > 1 ? count - 1 : 0) << 8) | ((uint32_t)poff << 16);
      e.pad = 0;
      lds_put(&ent[h], e);
    }
    n_ent++;
    W::sync();
    return (int)h;
  }

  // Returns 0 when the problem is of the shape this engine solves, a reason code otherwise. topo: on behalf of the spread engine
  // (topo_engine.h) — topology groups are its business, and the dictionary keys they spread over are variable keys too.
  KS_COLD int setup(bool topo = false) {
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const Dict& d = P.dict;
    const int nk = d.n_keys, iw = P.it_words, nr = P.n_res, n_its = P.n_its, nc = P.n_classes, T = P.n_templates;
    const ProblemView& Pv = P;
    if (!(topo ? P.plain_topo : P.plain) || P.n_rows != P.n_pods || nr > 4 || T > 32 || iw > kMaxItWords) return 1;
    // (instance types may use any operator: with positive sets on the claim side the NotIn / DoesNotExist escape of
    // requirements.go:260-265 never applies, so compatible() stays monotone)
    // templates: only In sets
    if (W::reduce_or(T, [&](int t) { return (uint64_t)(Pv.tmpl_reqs.complement[t] | (Pv.tmpl_reqs.has_gte ? Pv.tmpl_reqs.has_gte[t] : 0) | (Pv.tmpl_reqs.has_lte ? Pv.tmpl_reqs.has_lte[t] : 0)); })) return 3;
    // classes: only In sets; the keys they define are the variable keys
    if (W::reduce_or(nc, [&](int c) { return (uint64_t)Pv.cls_reqs.complement[c]; })) return 4;
    uint32_t vk = (uint32_t)W::reduce_or(nc, [&](int c) { return (uint64_t)Pv.cls_reqs.defined[c]; });
    if (topo) vk |= (uint32_t)W::reduce_or(P.topo.n_groups, [&](int g) { return Pv.topo.key[g] >= 0 ? (uint64_t)1 << Pv.topo.key[g] : (uint64_t)0; });
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
      return badc;
    });
    if (bad) return 8;
    W::for_n(kFastEnt, [&](int i) { ent[i].info = 0; ent[i].vmask = ~0ull; });   // (no requirement set is all ones: the top byte is a template id < 32)
    return 0;
  }

  // CanAdd (nodeclaim.go:124-242) of a claim in state `st` for the classes in the 64 slots of row j, one lane each -> the claim's
  // acceptance word of the row. A requirement set met for the first time gets its cache entry here. bail_code != 0: stop.
  KS_DEV uint64_t row_accepts(const FastClaim& st, int j) {
    const int t = (int)(st.vmask >> 56);
    const KS_LDS uint32_t* sc = hs->scls[j];
    uint64_t accm = 0;
    uint64_t todo = W::ballot([&](int l) { return sc[l] != kFastFree; });
    n_cold_tests += (unsigned long long)popc64(todo);
    while (todo) {
      LaneVar<uint64_t> missv;
      const uint64_t td = todo;
      uint64_t okb = 0, miss = 0, d0, d1;
      W::ballot4([&](int l) {
        missv.at(l) = 0;
        if (!((td >> l) & 1)) return 0;
        const FastSlot s = lds_get(&aslot[j * 64 + l]);
        if (!((s.tmplok >> t) & 1u)) return 0;
        const uint64_t m = st.vmask & s.cvmask;
        if (!fast_fields_ok(m, s.dmask)) return 0;
        FastEnt e;
        if (fast_lookup(ent, m, e) < 0) { missv.at(l) = m; return 2; }
        return fast_fits(pool, e, st.req, s.size) ? 1 : 0;
      }, okb, miss, d0, d1);
      accm |= okb;
      todo = miss;
      if (miss && create_entry(missv.bcast(ctz64(miss))) < 0) { bail_code = 20; return 0; }
    }
    return accm;
  }
  // every acceptance word of claim x, from its state (the loop met a requirement set that is not cached yet, a cache slot
  // that is not the set's first probe, or further Pareto vectors, while it refreshed the claim after a commit)
  KS_COLD int refresh_claim(int x) {
    x = (int)W::uniform((uint64_t)(uint32_t)x);
    const FastClaim st = cst.state((uint32_t)x);
    for (int j = 0; j < R; ++j) {
      const uint64_t a = row_accepts(st, j);
      if (bail_code) return -1;
      if (W::leader()) cst.put_acc((uint32_t)x, j, a);
    }
    W::sync();
    return 0;
  }

  // A class seen for the first time (or again after its slot was taken) gets a slot, and bit `slot` of every claim's
  // acceptance word: the class against all in-flight claims, one lane each. Returns the slot, bit 16: every slot was taken
  // and all of them were dropped (their classes take new slots when they come back), -1: stop (bail_code).
  KS_COLD int new_slot(int k) {
    k = (int)W::uniform((uint64_t)(uint32_t)k);
    int evicted = 0, slot = -1;
    // With several rows of class slots a class goes to the row that holds the most classes it could share a NodeClaim with
    // (templates both tolerate, a common value on every key both select on), the emptiest row among equals: classes that exclude
    // each other — pods pinned to different NodePools, BASELINE configs[3] — end up in different rows, and the refresh after a
    // commit skips every row none of whose classes the claim can still accept (fast_hot_run).
    int pref = 0;
    if constexpr (R > 1) {
      const FastSlot mine = Fk->cls[k];
      int best = -1;
      for (int j = 0; j < R; ++j) {
        const KS_LDS uint32_t* sc = hs->scls[j];
        const KS_LDS FastSlot* as = aslot + j * 64;
        const uint64_t used = W::ballot([&](int l) { return sc[l] != kFastFree; });
        if (used == ~0ull) continue;
        const uint64_t friends = W::ballot([&](int l) {
          if (sc[l] == kFastFree) return false;
          const FastSlot o = lds_get(&as[l]);
          return (o.tmplok & mine.tmplok) != 0 && fast_fields_ok(o.cvmask & mine.cvmask, o.dmask & mine.dmask);
        });
        const int score = popc64(friends) * 128 + (64 - popc64(used));
        if (score > best) { best = score; pref = j; }
      }
    }
    for (int jj = 0; jj < R && slot < 0; ++jj) {
      const int j = (jj + pref) % R;
      const KS_LDS uint32_t* sc = hs->scls[j];
      const uint64_t fr = W::ballot([&](int l) { return sc[l] == kFastFree; });
      if (fr) slot = j * 64 + ctz64(fr);
    }
    if (slot < 0) {
      // every slot taken by a class that has entries left: forget them all
      KS_LDS FastHot* h = hs;
      W::each([&](int l) { for (int j = 0; j < R; ++j) { h->scls[j][l] = kFastFree; h->cur[j][l] = 0; } });
      W::sync();
      n_evict++;
      evicted = 1 << 16;
      slot = 0;
    }
    const int row = slot >> 6, sl = slot & 63;
    const FastSlot rec = Fk->cls[k];
    if (W::leader()) { lds_put(&aslot[slot], rec); hs->scls[row][sl] = (uint32_t)k; hs->cur[row][sl] = 0; }
    W::sync();
    const uint64_t bit = 1ull << sl;
    const int nc = n_claims;
    const typename FastMem<GS, R>::States cs_ = cst;
    for (int x0 = 0; x0 < nc; x0 += 64) {
      // the plain verdicts first, straight-line (the claim's record, the first probe of the requirement-set cache, one compare per
      // predicate); the lanes that need the long way — a set that is not at its first probe or not cached yet, an entry with further
      // Pareto vectors — go through the probe loop below
      LaneVar<uint64_t> mlv, evm, vmv;
      LaneVar<int32_t> c0, c1, c2, c3, r0, r1, r2, r3;
      W::each([&](int l) {
        const int x = x0 + l < nc ? x0 + l : nc - 1;
        const FastClaim st = cs_.state((uint32_t)x);
        const uint64_t m = st.vmask & rec.cvmask;
        const FastEnt e = lds_get16(&ent[fast_hash(m)]);
        mlv.at(l) = m; evm.at(l) = e.vmask; vmv.at(l) = st.vmask;
        c0.at(l) = e.cap[0]; c1.at(l) = e.cap[1]; c2.at(l) = e.cap[2]; c3.at(l) = e.cap[3];
        r0.at(l) = st.req[0]; r1.at(l) = st.req[1]; r2.at(l) = st.req[2]; r3.at(l) = st.req[3];
      });
      const uint64_t validm = W::ballot([&](int l) { return x0 + l < nc; });
      const uint64_t basem = validm & W::ballot([&](int l) { return ((rec.tmplok >> (vmv.at(l) >> 56)) & 1u) != 0 && fast_fields_ok(mlv.at(l), rec.dmask); });
      const uint64_t simm = W::ballot([&](int l) { return evm.at(l) == mlv.at(l); });
      const uint64_t fitm = W::ballot([&](int l) { return (rec.size[0] <= c0.at(l) - r0.at(l)) & (rec.size[1] <= c1.at(l) - r1.at(l)) & (rec.size[2] <= c2.at(l) - r2.at(l)) & (rec.size[3] <= c3.at(l) - r3.at(l)); });
      uint64_t accm = basem & simm & fitm;
      uint64_t todo = basem & ~simm;
      n_cold_tests += (unsigned long long)popc64(validm);
      while (todo) {
        LaneVar<uint64_t> missv;
        const uint64_t td = todo;
        uint64_t okb = 0, miss = 0, d0, d1;
        W::ballot4([&](int l) {
          missv.at(l) = 0;
          if (!((td >> l) & 1)) return 0;
          const FastClaim st = cs_.state((uint32_t)(x0 + l));
          if (!((rec.tmplok >> (st.vmask >> 56)) & 1u)) return 0;
          const uint64_t m = st.vmask & rec.cvmask;
          if (!fast_fields_ok(m, rec.dmask)) return 0;
          FastEnt e;
          if (fast_lookup(ent, m, e) < 0) { missv.at(l) = m; return 2; }
          return fast_fits(pool, e, st.req, rec.size) ? 1 : 0;
        }, okb, miss, d0, d1);
        accm |= okb;
        todo = miss;
        if (miss && create_entry(missv.bcast(ctz64(miss))) < 0) { bail_code = 20; return -1; }
      }
      W::each([&](int l) {
        if (x0 + l < nc) {
          const uint64_t w = cs_.acc((uint32_t)(x0 + l), row);
          cs_.put_acc((uint32_t)(x0 + l), row, (w & ~bit) | (((accm >> l) & 1) ? bit : 0ull));
        }
      });
    }
    W::sync();
    return slot | evicted;
  }

  // any path of pdqsort other than the single stable move: run the emulation, compare the order before and after
  // (lo_..hi_ = the positions whose claim changed, hi_ < lo_: none)
  KS_COLD void slow_sort(int n, int defect, int app) {
    order.n = (int)W::uniform((uint64_t)(uint32_t)n); order.defect = (int)W::uniform((uint64_t)(uint32_t)defect); order.defect_append = W::uniform((uint64_t)app) != 0;
    n = order.n;
    const o16 oo = order.ord; const typename FastMem<GS, R>::s16 sn = snap;
    if constexpr (FastMem<GS, R>::kSnapHbm) {
      W::copy8(sn, oo, n);
      order.sort();
      // (a snapshot in HBM: eight rounds of loads in flight per search step)
      lo_ = W::find_first8(0, n, [&](int i) { return sn[i] != oo[i]; });
      if (lo_ >= n) { lo_ = 0; hi_ = -1; return; }
      hi_ = W::find_last8(0, n, [&](int i) { return sn[i] != oo[i]; });
    } else {
      // Both arrays in LDS, 16-byte aligned: the snapshot and the two searches in pieces of eight claim ids (one ds_read_b128 per
      // lane: 512 positions per step instead of 64 — the copy and the compares were half of a slow sort's 27k cycles). The
      // ids past n in the last piece are copied along and stay equal: the sort does not touch them.
      const int n8 = (n + 7) >> 3;
      const KS_LDS u32x4_alias* const o8 = (const KS_LDS u32x4_alias*)oo;
      KS_LDS u32x4_alias* const s8 = (KS_LDS u32x4_alias*)sn;
      W::for_n(n8, [&](int i) { s8[i] = o8[i]; });
      order.sort();
      auto differs = [&](int i) { const u32x4_alias a = s8[i], b = o8[i]; return ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) != 0; };
      const int c0 = W::find_first(0, n8, differs);
      if (c0 >= n8) { lo_ = 0; hi_ = -1; return; }
      const int c1 = W::find_last(0, n8, differs);
      lo_ = W::find_first(c0 * 8, c0 * 8 + 8 < n ? c0 * 8 + 8 : n, [&](int i) { return sn[i] != oo[i]; });
      hi_ = W::find_last(c1 * 8, c1 * 8 + 8 < n ? c1 * 8 + 8 : n, [&](int i) { return sn[i] != oo[i]; });
    }
  }
  // The new claim (appended at n-1 with one pod) takes its place behind the last claim with at most one pod. Returns its
  // position b >= 0 (Mp->acc = the class slots that accept it, from new_claim), -1 when pdqsort left the single-move path (lo_/hi_).
  // arr[b+1 .. a] = arr[b .. a-1] for an LDS-resident array of 16-bit entries, 512 entries per step: every lane takes a 16-byte
  // piece (eight entries), shifts it by one entry through its registers (the entry that crosses into the piece comes from the lane
  // below it) and writes it back in place — from the top piece down, so that a step reads nothing an earlier step wrote. The
  // one-entry-per-lane move this replaces cost a new claim's sort.Slice 43 dependent read-write rounds at 2,763 claims.
  KS_DEV void shift_right16(KS_LDS uint16_t* arr, int b, int a) {
    KS_LDS u32x4_alias* const v = (KS_LDS u32x4_alias*)arr;
    const KS_LDS uint32_t* const dw = (const KS_LDS uint32_t*)arr;
    for (int e0 = a & ~511; e0 >= 0 && e0 + 511 >= b; e0 -= 512) {
      LaneVar<uint32_t> d0, d1, d2, d3, pv;
      const uint32_t below = e0 > 0 ? dw[(e0 >> 1) - 1] : 0u;      // the two entries under the step's first piece
      W::each([&](int l) {
        const int e = e0 + 8 * l;
        d0.at(l) = 0; d1.at(l) = 0; d2.at(l) = 0; d3.at(l) = 0;
        if (e <= a && e + 7 >= b) { const u32x4_alias x = v[(e0 >> 3) + l]; d0.at(l) = x[0]; d1.at(l) = x[1]; d2.at(l) = x[2]; d3.at(l) = x[3]; }
      });
      // (the exchange runs on EVERY lane, then lane 0 takes the word below the step: a lane that is switched off during ds_bpermute
      // reads as zero for its neighbour — the first build had the exchange inside the select and lane 1 got a zero from lane 0)
      LaneVar<uint32_t> up;
      W::each([&](int l) { up.at(l) = d3.shuffle(l, (l + 63) & 63); });
      W::each([&](int l) { pv.at(l) = l ? up.at(l) : below; });
      W::each([&](int l) {
        const int e = e0 + 8 * l;
        if (!(e <= a && e + 7 > b)) return;
        auto mix = [&](uint32_t cur, uint32_t prev, int elo) {   // entries elo (low half) and elo + 1 (high half) of one word
          const uint32_t sh = (cur << 16) | (prev >> 16);
          const uint32_t m = ((elo > b && elo <= a) ? 0xFFFFu : 0u) | ((elo + 1 > b && elo + 1 <= a) ? 0xFFFF0000u : 0u);
          return (sh & m) | (cur & ~m);
        };
        u32x4_alias y;
        y[0] = mix(d0.at(l), pv.at(l), e); y[1] = mix(d1.at(l), d0.at(l), e + 2); y[2] = mix(d2.at(l), d1.at(l), e + 4); y[3] = mix(d3.at(l), d2.at(l), e + 6);
        v[(e0 >> 3) + l] = y;
      });
      W::sync();
    }
  }
  KS_COLD int place_new_claim(int n) {
    n = (int)W::uniform((uint64_t)(uint32_t)n);
    const int a = n - 1;
    const bool exact = n <= 12 || (n >= 50 && !fast_sampled(n, a));
    if (!exact) { slow_sort(n, a, 1); return -1; }
    if constexpr (!FastMem<GS, R>::kOrderHbm) {
      // insertion sort (n <= 12) and partialInsertionSort (pdq_emul.h) both make ONE stable move of the appended claim: left,
      // behind the last claim whose count is not larger than its own — one pod, the smallest count there is: behind the claims
      // with one pod, which the sorted array keeps in front
      const o16 kq = order.key, oq = order.ord;
      const int b = W::find_first(0, a, [&](int i) { return kq[i] > 1u; });
      if (b < a) {
        const uint32_t mo = oq[a];
        shift_right16(kq, b, a);
        shift_right16(oq, b, a);
        if (W::leader()) { kq[b] = (uint16_t)1; oq[b] = (uint16_t)mo; }
        W::sync();
      }
      return b;
    } else {
      order.n = n; order.defect = a; order.defect_append = true;
      order.sort();
      const o16 kq = order.key;
      return W::find_first(0, n, [&](int i) { return kq[i] > 1u; }) - 1;   // the claims with one pod are the prefix it joined the end of
    }
  }

  // addToNewNodeClaim (scheduler.go:695-790) for a pod no in-flight claim accepted: 1 = claim n created (appended to the
  // order with one pod; its acceptance words computed), 0 = the engine must stop (bail_code; -1 = capacity).
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
      FastClaim ns;
      ns.vmask = m;
      for (int q = 0; q < 4; ++q) ns.req[q] = cs.size[q];
      if (W::leader()) {
        cst.put_state((uint32_t)c, ns);
        F.c_hostseq[c] = host_seq;
        order.key[n] = 1; order.ord[n] = (uint16_t)c;   // order.append
      }
      W::sync();
      for (int j = 0; j < R; ++j) {
        const uint64_t aw = row_accepts(ns, j);
        if (bail_code) return 0;
        if (W::leader()) { cst.put_acc((uint32_t)c, j, aw); Mp->acc[j] = aw; }
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
      const typename FastMem<GS, R>::States ls = cst;
      const KS_LDS FastEnt* en = ent;
      W::for_n(n, [&](int i) { const uint32_t c = oo[i]; go[i] = c; gn[c] = ok_[i]; });
      W::for_n(n, [&](int c) {
        const FastClaim st = ls.state((uint32_t)c);
        FastEnt e;
        gs[c] = st;
        ge[c] = (uint16_t)fast_lookup(en, st.vmask, e);
      });
      W::store(S.n_claims_out, n_claims);
    }
    if (status) W::store(S.status_out, status);
    Counters c{};
    c.bin_evaluations = n_tests + n_cold_tests; c.full_evaluations = n_steps; c.queue_pops = steps; c.sorts = steps; c.slow_sorts = order.slow_sorts;
    c.column_resets = (unsigned long long)n_evict; c.ref_bin_evaluations = n_ref + n_ref_extra;
    c.cycles[20] = (unsigned long long)(bail_code > 0 ? bail_code : 0);
    if (tc) for (int i = 0; i < 16; ++i) c.cycles[i] = tc[i];
#ifdef KSOLVE_PHASE_TIMERS
    for (int i = 0; i < 4; ++i) c.cycles[16 + i] = hs->hw[i];
    c.cycles[18] = (hs->hw[2] << 32) | (unsigned long long)(hs->mail.simd[0] | (hs->mail.simd[1] << 8) | (hs->mail.simd[2] << 16) | (hs->mail.simd[3] << 24));
#endif
    if (W::leader()) *S.counters = c;
    W::sync();
  }
};

template <int GS, int R>
struct FastHotCtx {   // LDS pointers of the loop, passed by value
  typename FastMem<GS, R>::o16 okey, oord; typename FastMem<GS, R>::States cst; KS_LDS FastEnt* ent; KS_LDS int32_t* pool;
  KS_LDS FastSlot* aslot; KS_LDS FastHot* hs;
};
enum { FEV_DONE = 0, FEV_REFRESH = 1, FEV_SLOT = 2, FEV_SLOWSORT = 3, FEV_PLACE = 4, FEV_NEWCLAIM = 5, FEV_COUNT = 6, FEV_SLOW = 7, FEV_CONT = 8, FEV_DEAD = 9 };

// ---- the refresher (second wavefront of ksolve_pack_fast2) ---------------------------------------------------------------
// The classes of the slots, one lane each, as the refresh needs them (registers of the refresher between two changes of the slots)
template <int R>
struct FastHelperK {
  LaneVar<uint32_t> tok[R];
  LaneVar<uint64_t> cvm[R], dm[R], gd[R];
  LaneVar<int32_t> z0[R], z1[R], z2[R], z3[R];
  uint32_t gen = 0xFFFFFFFFu, miss = 0;
};
template <class W, int R>
KS_FN void fast_helper_consts(FastHelperK<R>& k, const KS_LDS FastSlot* aslot, const KS_LDS FastHot* hs) {
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const FastSlot s = lds_get(&aslot[j * 64 + l]);
      k.cvm[j].at(l) = s.cvmask; k.dm[j].at(l) = s.dmask; k.gd[j].at(l) = (s.dmask << 1) & ~s.dmask;
      k.z0[j].at(l) = s.size[0]; k.z1[j].at(l) = s.size[1]; k.z2[j].at(l) = s.size[2]; k.z3[j].at(l) = s.size[3];
      k.tok[j].at(l) = hs->scls[j][l] == kFastFree ? 0u : s.tmplok;   // (a slot whose class left inside the placer's run keeps its bits: nobody reads them before the next activation)
    }
  });
}
// Request s: CanAdd (nodeclaim.go:124-242) of the claim as it stands after the add, for the classes of all slots (lane = slot) — the
// same predicates as the one-wavefront loop's refresh (fast_hot_run<.., false>)
template <class W, int GS, int R>
KS_FN void fast_helper_serve(FastHelperK<R>& k, FastHotCtx<GS, R> cx, uint32_t s_, uint32_t w_) {
  KS_LDS FastHot* const hs = cx.hs;
  KS_LDS FastMail* const mail = &hs->mail;
  W::order();
  const uint32_t gen = (uint32_t)fast_uniform((int)mail->gen);
  if (gen != k.gen) { fast_helper_consts<W, R>(k, cx.aslot, hs); k.gen = gen; k.miss = 0; }
  const uint32_t x = (uint32_t)fast_uniform((int)(w_ & 0xFFFFu));
  const FastClaim st = cx.cst.state(x);      // (the placer does not touch a claim with a request in flight)
  const uint64_t vmask = W::uniform(st.vmask);
  const int32_t req[4] = {fast_uniform(st.req[0]), fast_uniform(st.req[1]), fast_uniform(st.req[2]), fast_uniform(st.req[3])};
  const uint32_t tbit = 1u << (uint32_t)(vmask >> 56);
  LaneVar<uint64_t> mlv[R], evm[R];
  LaneVar<int32_t> c0[R], c1[R], c2[R], c3[R];
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint64_t m = vmask & k.cvm[j].at(l);
      const FastEnt e = ent_live(&cx.ent[fast_hash(m)]);
      mlv[j].at(l) = m; evm[j].at(l) = e.vmask;
      c0[j].at(l) = e.cap[0]; c1[j].at(l) = e.cap[1]; c2[j].at(l) = e.cap[2]; c3[j].at(l) = e.cap[3];
    }
  });
  uint64_t accw[R];
  bool miss = false;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint64_t tokm = W::ballot([&](int l) { return (k.tok[j].at(l) & tbit) != 0; });
    const uint64_t fldm = W::ballot([&](int l) { return (((mlv[j].at(l) & k.dm[j].at(l)) + k.dm[j].at(l)) & k.gd[j].at(l)) == k.gd[j].at(l); });
    const uint64_t simm = W::ballot([&](int l) { return evm[j].at(l) == mlv[j].at(l); });
    const uint64_t f0 = W::ballot([&](int l) { return k.z0[j].at(l) <= c0[j].at(l) - req[0]; });
    const uint64_t f1 = W::ballot([&](int l) { return k.z1[j].at(l) <= c1[j].at(l) - req[1]; });
    const uint64_t f2 = W::ballot([&](int l) { return k.z2[j].at(l) <= c2[j].at(l) - req[2]; });
    const uint64_t f3 = W::ballot([&](int l) { return k.z3[j].at(l) <= c3[j].at(l) - req[3]; });
    const uint64_t basem = tokm & fldm, fitm = f0 & f1 & f2 & f3;
    uint64_t accm = basem & simm & fitm;
    const uint64_t oddm = basem & ~simm;   // set not at its first probe / not cached / an entry with further Pareto vectors (kFastExtBit)
    if (KS_UNLIKELY(oddm != 0)) {
      uint64_t ok2 = 0, missm = 0;
      W::ballot2([&](int l) {
        if (!((oddm >> l) & 1)) return 0;
        FastEnt e;
        if (fast_lookup(cx.ent, mlv[j].at(l), e) < 0) return 2;
        const int32_t sz[4] = {k.z0[j].at(l), k.z1[j].at(l), k.z2[j].at(l), k.z3[j].at(l)};
        return fast_fits(cx.pool, e, req, sz) ? 1 : 0;
      }, ok2, missm);
      accm |= ok2;
      if (missm) miss = true;
    }
    accw[j] = accm;
  }
  if (W::leader()) {
    if (KS_UNLIKELY(miss)) mail->miss_x[s_ & 1u] = (int32_t)x;
    else {
#pragma unroll
      for (int j = 0; j < R; ++j) cx.cst.put_acc(x, j, accw[j]);
    }
  }
  if (miss) k.miss = 1u;
  W::order();
  if (W::leader()) mail_store(&mail->done, (s_ << 1) | k.miss);
}
// before the two wavefronts part (the kernel's first statement, in front of its only barrier)
KS_FN void fast_mail_init(KS_LDS FastMail* mail) {
  mail->word[0] = 0; mail->word[1] = 0; mail->done = 0; mail->gen = 0; mail->quit = 0; mail->miss_x[0] = -1; mail->miss_x[1] = -1;
}
#if !KS_DEVICE
// the emulation's refresher: a function the placer calls when it waits for a request (or, KSOLVE_EMU_REFRESHER_EAGER=1, right after
// posting one) — the slowest and the fastest refresher there can be; the device's is anywhere in between
template <int R> inline FastHelperK<R>& fast_emu_helper_k() { static thread_local FastHelperK<R> k; return k; }
inline bool& fast_emu_helper_eager() { static thread_local bool e = false; return e; }   // (set per solve by the emulation's launcher)
#endif
#if KS_DEVICE
// the refresher's life: poll, serve in order, leave when the placer says so
template <class W, int GS, int R>
KS_DEV void fast_helper_run(const FastWork* f, char* lds) {
  const FastPlan& pl = f->plan;
  FastHotCtx<GS, R> cx;
  cx.hs = fast_uniform((KS_LDS FastHot*)(lds + pl.off_hot));
  cx.cst.p = fast_uniform((KS_LDS FastRec<R>*)(lds + pl.off_state));
  cx.okey = nullptr; cx.oord = nullptr;
  cx.ent = fast_uniform((KS_LDS FastEnt*)(lds + pl.off_ent));
  cx.pool = fast_uniform((KS_LDS int32_t*)(lds + pl.off_pool));
  cx.aslot = fast_uniform((KS_LDS FastSlot*)(lds + pl.off_slot));
  KS_LDS FastMail* const mail = &cx.hs->mail;
  FastHelperK<R> k;
  uint32_t seen = 0;
  for (;;) {
    const uint32_t w0 = mail_load(&mail->word[0]), w1 = mail_load(&mail->word[1]), quit = mail_load(&mail->quit);   // (the reads in flight together)
    if (fast_uniform((int)quit) != 0) break;
    const uint32_t nxt = seen + 1u;
    const uint32_t w = (uint32_t)fast_uniform((int)((nxt & 1u) ? w1 : w0));
    if ((w >> 16) != (nxt & 0xFFFFu)) continue;
    fast_helper_serve<W, GS, R>(k, cx, nxt, w);
    seen = nxt;
  }
}
#endif

// The loop that places pods: a function of its own, WITHOUT calls — whatever happens rarely (a requirement set seen for the
// first time, a class without a slot, a new claim, pdqsort leaving its single-move path) ends the run with an event code; the
// driver handles it through FastCold and runs the loop again. So the compiler allocates registers for this loop alone.
//
// One step of Solve() (scheduler.go:440-519) for the pod at the head of the queue, class k in slot s:
//   select   addToInflightNode's "lowest index that accepts" (scheduler.go:667-686) is the first position at or after the
//            class's cursor whose claim has bit s of its acceptance word set: 64 positions per step, one lane each, two
//            dependent reads (order -> claim record), no test — the words are exact;
//   commit   NodeClaim.Add (nodeclaim.go:247-263) on that claim, and the stable move that the sort.Slice of the next add
//            (scheduler.go:598) makes of it, from the counts the select step already holds in registers;
//   refresh  the claim's acceptance words from its new state: CanAdd for the classes of all slots at once, lane = slot, the
//            classes' records in registers, one requirement-set cache read per lane.
// Nothing is speculated, so nothing is validated: ~110 instructions per pod, three dependent LDS round trips.
template <class W, int GS, int R>
KS_COLD int fast_slow_run(FastHotCtx<GS, R> cx, int budget) {
  budget = fast_uniform(budget);
  typedef typename FastMem<GS, R>::o16 o16;
  const unsigned long long t_in = W::clock();
  const o16 okey = fast_uniform(cx.okey), oord = fast_uniform(cx.oord);
  const typename FastMem<GS, R>::States cst = fast_uniform(cx.cst);
  KS_LDS FastEnt* const ent = fast_uniform(cx.ent);
  KS_LDS int32_t* const pool = fast_uniform(cx.pool);
  KS_LDS FastSlot* const aslot = fast_uniform(cx.aslot);
  KS_LDS FastHot* const hs = fast_uniform(cx.hs);
  // ---- state in ----
  const int np = fast_uniform(hs->np), max_steps_in = fast_uniform(hs->max_steps);
  const int max_steps = max_steps_in < 0 ? 0x7FFFFFFF : max_steps_in;   // (no limit: one compare per pod)
  const KS_GLOBAL uint32_t* const gqcls = (const KS_GLOBAL uint32_t*)fast_uniform(hs->q_class);
  const volatile int* const cancel = fast_uniform(hs->cancel);
  KS_GLOBAL uint32_t* const gqclaim = (KS_GLOBAL uint32_t*)fast_uniform(hs->q_claim);
  KS_GLOBAL uint32_t* const gqcnt = (KS_GLOBAL uint32_t*)fast_uniform(hs->q_cnt);
  int base = fast_uniform(hs->base), bi = fast_uniform(hs->bi), bn = fast_uniform(hs->bn), n = fast_uniform(hs->n), steps = fast_uniform(hs->steps), status = fast_uniform(hs->status);
  int pend_a = fast_uniform(hs->pend_a), pend_x = fast_uniform(hs->pend_x); uint32_t pend_mv = (uint32_t)fast_uniform((int)hs->pend_mv); bool pend_new = fast_uniform(hs->pend_new) != 0;
  unsigned long long n_steps = W::uniform(hs->n_steps), n_ref = W::uniform(hs->n_ref);   // n_steps: select steps beyond a pod's first
  LaneVar<uint32_t> cur[R], scls[R], tok[R], nxt_cls, bcls, oclaim, ocnt;
  LaneVar<uint64_t> cvm[R], dm[R], gd[R];                  // the class of lane's slot: values it admits, fields it selects on, their guard bits
  LaneVar<int32_t> z0[R], z1[R], z2[R], z3[R];             // its requests
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      cur[j].at(l) = hs->cur[j][l]; scls[j].at(l) = hs->scls[j][l];
      const FastSlot s = lds_get(&aslot[j * 64 + l]);
      cvm[j].at(l) = s.cvmask; dm[j].at(l) = s.dmask; gd[j].at(l) = (s.dmask << 1) & ~s.dmask;
      z0[j].at(l) = s.size[0]; z1[j].at(l) = s.size[1]; z2[j].at(l) = s.size[2]; z3[j].at(l) = s.size[3];
      tok[j].at(l) = scls[j].at(l) == kFastFree ? 0u : s.tmplok;   // a free slot accepts nothing
    }
    nxt_cls.at(l) = hs->nxt_cls[l]; bcls.at(l) = hs->bcls[l]; oclaim.at(l) = hs->oclaim[l]; ocnt.at(l) = hs->ocnt[l];
  });
  int ev = FEV_DONE, ev_arg = 0;
  // choosePivot's sampled positions (fast_sampled) as three starts; n is fixed inside one run of this function. With n <= 12 every
  // re-sort is the stable insertion sort; with 12 < n < 50 every re-sort that has something to move is pdqsort's other path.
  const bool always_exact = n <= 12, never_exact = n > 12 && n < 50;
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
      bi = 0;
      const int nb = base + 64;
      W::each([&](int l) { bcls.at(l) = nxt_cls.at(l); });
      W::each([&](int l) { if (nb + l < np) nxt_cls.at(l) = gqcls[nb + l]; });
      if ((base & 1023) == 0 && cancel) {
        // > 0: ksolve_cancel / the deadline. < 0 (tests only, KSOLVE_TEST_CANCEL_AT): as if the cancel landed once -flag pods were placed
        const int cv = fast_uniform((int)W::poll_flag(cancel));
        if (cv > 0 || (cv < 0 && base >= -cv)) { status = 2; bn = 0; break; }
      }
    }
    if (KS_UNLIKELY(steps >= max_steps)) { status = 2; break; }
    // ---- sort.Slice (scheduler.go:598) for a move the last commit left behind ----
    if (KS_UNLIKELY(pend_a >= 0)) {
      if (pend_new) { ev = FEV_PLACE; break; }
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
          for (int j = 0; j < R; ++j) { const uint32_t r = cur[j].at(l); cur[j].at(l) = r - (uint32_t)(((uint32_t)a < r && r <= (uint32_t)b) ? 1 : 0); }
        });
      }
    }
    KS_SEC(ts0)   // block fetch, pending move
    // ---- the pod's class and its slot ----
    const uint32_t clsw = bcls.bcast(bi), kcls = clsw & ~kFastLastBit;
    int row = -1, sl = 0;
    if constexpr (R == 1) {
      const uint64_t m0 = W::ballot([&](int l) { return scls[0].at(l) == kcls; });
      if (KS_LIKELY(m0 != 0)) { row = 0; sl = ctz64(m0); }
    } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const uint64_t mj = W::ballot([&](int l) { return scls[j].at(l) == kcls; });
        if (mj != 0 && row < 0) { row = j; sl = ctz64(mj); }
      }
    }
    if (KS_UNLIKELY(row < 0)) { ev = FEV_SLOT; ev_arg = (int)kcls; break; }
    if constexpr (R == 1) row = 0;
    const int slot = row * 64 + sl;
    const FastSlot cs = lds_get(&aslot[slot]);
    uint32_t rc0 = cur[0].bcast(sl);
#pragma unroll
    for (int j = 1; j < R; ++j) { const uint32_t cj = cur[j].bcast(sl); rc0 = row == j ? cj : rc0; }
    // ---- select: addToInflightNode (scheduler.go:658-692), positions r .. r+63, one lane each: the order's entry, then the
    // claim's whole record (state and acceptance word in one round trip: the commit needs the state of the claim it finds) ----
    LaneVar<uint64_t> mvv;
    LaneVar<uint32_t> xv, kv;
    LaneVar<int32_t> q0, q1, q2, q3;
    uint32_t r = rc0, r0 = rc0;
    uint64_t okm = 0;
    const int nm1 = n - 1;
    const uint64_t slbit = 1ull << sl;
    auto scan = [&](uint32_t rr0) {
      return W::ballot([&](int l) {
        const int p = (int)rr0 + l;
        const int pc = p < nm1 ? p : nm1;                 // clamped: no lane is switched off for the reads
        const uint64_t want = p < n ? slbit : 0ull;       // (ready before the reads come back: the ballot is one AND and one compare behind them)
        const uint32_t x = oord[pc], k = okey[pc];
        const FastClaim st = cst.state(x);
        const uint64_t aw = cst.acc(x, row);
        xv.at(l) = x; kv.at(l) = p < n ? k : 0xFFFFFFFFu;
        q0.at(l) = st.req[0]; q1.at(l) = st.req[1]; q2.at(l) = st.req[2]; q3.at(l) = st.req[3];
        mvv.at(l) = st.vmask;
        return (aw & want) != 0;
      });
    };
    if (KS_LIKELY((int)r < n)) {
      okm = scan(r0);
      if (KS_UNLIKELY(okm == 0)) {
        // not among the 64 claims at the cursor: the rest of the order, 64 positions per step
        r = (uint32_t)((int)r0 + 64 < n ? (int)r0 + 64 : n);
        while ((int)r < n) {
          r0 = r;
          okm = scan(r0);
          n_steps++;
          if (okm != 0) break;
          r = (uint32_t)((int)r0 + 64 < n ? (int)r0 + 64 : n);
        }
      }
    }
    KS_SEC(ts1)   // select
    if (KS_UNLIKELY(okm == 0)) {
      // no in-flight claim accepts the pod: addToNewNodeClaim; the driver moves on to the next pod
      W::each([&](int l) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const bool me = j * 64 + l == slot;
          cur[j].at(l) = me ? r : cur[j].at(l);
          if (clsw & kFastLastBit) { scls[j].at(l) = me ? kFastFree : scls[j].at(l); tok[j].at(l) = me ? 0u : tok[j].at(l); }
        }
      });
      steps++;
      ev = FEV_NEWCLAIM; ev_arg = slot;
      break;
    }
    // ---- commit: NodeClaim.Add (nodeclaim.go:247-263) ----
    const int first_ok = ctz64(okm);
    const int a = (int)r0 + first_ok;
    const int x = (int)xv.bcast(first_ok);
    const uint32_t cnt = kv.bcast(first_ok);
    if (KS_UNLIKELY(cnt >= 65534u)) { ev = FEV_COUNT; break; }
    FastClaim ns;
    ns.vmask = mvv.bcast(first_ok) & cs.cvmask;
    ns.req[0] = q0.bcast(first_ok) + cs.size[0]; ns.req[1] = q1.bcast(first_ok) + cs.size[1];
    ns.req[2] = q2.bcast(first_ok) + cs.size[2]; ns.req[3] = q3.bcast(first_ok) + cs.size[3];
    n_ref += (unsigned long long)a + 1;
    // The sort.Slice of the NEXT add (scheduler.go:598) repairs this claim's position: one stable move past the claims
    // with a smaller count. When the next add follows inside this block and those claims are all among the positions
    // just read, their counts and ids are in registers already: move now, without reading the order again.
    const uint32_t mvn = cnt + 1;
    int s_ = -1;   // claims the move passes; -1: not decided here (the pending path of the next add)
    if (KS_LIKELY(bi + 1 < bn && steps + 1 < max_steps)) {
      const uint64_t lessm = W::ballot([&](int l) { return kv.at(l) < mvn; });   // lanes past n hold 0xFFFFFFFF; the lanes up to first_ok are shifted out
      const uint64_t t = first_ok == 63 ? 0ull : (lessm >> (first_ok + 1));
      const int sm = t == ~0ull ? 64 : ctz64(~t);
      const bool in_window = first_ok + 1 + sm < 64 || (int)r0 + 64 >= n;
      if (KS_LIKELY(sm == 0)) {
        // no move at all (the next claim has at least the new count: sorted as it stands, pdqsort finds no descent)
        if (KS_LIKELY(in_window)) s_ = 0;
      } else if (in_window && (always_exact || (!never_exact && !(((uint32_t)a - e1 <= 2u) | ((uint32_t)a - e2 <= 2u) | ((uint32_t)a - e3 <= 2u))))) {
        // the single stable move: lanes first_ok+1 .. first_ok+sm step one position to the left, the claim lands behind them
        s_ = sm;
        const int rb = (int)r0;
        W::each([&](int l) { if (l > first_ok && l <= first_ok + sm) { okey[rb + l - 1] = (uint16_t)kv.at(l); oord[rb + l - 1] = (uint16_t)xv.at(l); } });
        const uint32_t ua1 = (uint32_t)a + 1u, su = (uint32_t)sm;
        W::each([&](int l) {
#pragma unroll
          for (int j = 0; j < R; ++j) { const uint32_t rr = cur[j].at(l); cur[j].at(l) = rr - (uint32_t)((rr - ua1) < su); }   // a < rr <= a + sm
        });
      }
    }
    if (KS_UNLIKELY(s_ < 0)) { pend_a = a; pend_x = x; pend_mv = mvn; }
    // the pod's result; the class's cursor comes to a (the claims between its old place and this one rejected the class for
    // good); its slot is free after the class's last entry (branch-free: kFastFree is all ones)
    {
      const uint32_t lastm = (clsw & kFastLastBit) ? 0xFFFFFFFFu : 0u;
      const int bb = bi;
      W::each([&](int l) {
        const bool mine = l == bb;
        oclaim.at(l) = mine ? (uint32_t)x : oclaim.at(l); ocnt.at(l) = mine ? cnt : ocnt.at(l);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const bool me = j * 64 + l == slot;
          const uint32_t fm = me ? lastm : 0u;
          cur[j].at(l) = me ? (uint32_t)a : cur[j].at(l);
          scls[j].at(l) |= fm; tok[j].at(l) &= ~fm;
        }
      });
    }
    KS_SEC(ts2)   // commit
    // ---- refresh: CanAdd (nodeclaim.go:124-242) of the claim as it stands now, for the classes of all slots (lane = slot) ----
    const uint32_t tbit = 1u << (uint32_t)(ns.vmask >> 56);
    bool cold_refresh = false;
    uint64_t accw[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      LaneVar<uint64_t> mlv, evm;
      LaneVar<int32_t> c0, c1, c2, c3;
      LaneVar<uint32_t> einfo;
      W::each([&](int l) {
        const uint64_t m = ns.vmask & cvm[j].at(l);
        const FastEnt e = lds_get(&ent[fast_hash(m)]);
        mlv.at(l) = m; evm.at(l) = e.vmask; einfo.at(l) = e.info;
        c0.at(l) = e.cap[0]; c1.at(l) = e.cap[1]; c2.at(l) = e.cap[2]; c3.at(l) = e.cap[3];
      });
      // every predicate is one compare whose result is the 64-lane mask; the masks are combined in scalar registers
      const uint64_t tokm = W::ballot([&](int l) { return (tok[j].at(l) & tbit) != 0; });                       // taints, custom keys (nodeclaim.go:126; requirements.go:185-193)
      const uint64_t fldm = W::ballot([&](int l) { return (((mlv.at(l) & dm[j].at(l)) + dm[j].at(l)) & gd[j].at(l)) == gd[j].at(l); });   // Compatible on the keys the class selects on
      const uint64_t simm = W::ballot([&](int l) { return evm.at(l) == mlv.at(l); });                          // the cache's first probe is this requirement set
      const uint64_t f0 = W::ballot([&](int l) { return z0[j].at(l) <= c0.at(l) - ns.req[0]; });
      const uint64_t f1 = W::ballot([&](int l) { return z1[j].at(l) <= c1.at(l) - ns.req[1]; });
      const uint64_t f2 = W::ballot([&](int l) { return z2[j].at(l) <= c2.at(l) - ns.req[2]; });
      const uint64_t f3 = W::ballot([&](int l) { return z3[j].at(l) <= c3.at(l) - ns.req[3]; });
      const uint64_t extm = W::ballot([&](int l) { return (einfo.at(l) & 0xFF00u) != 0; });                   // further Pareto vectors
      const uint64_t basem = tokm & fldm, fitm = f0 & f1 & f2 & f3;
      uint64_t accm = basem & simm & fitm;
      const uint64_t oddm = basem & (~simm | (~fitm & extm));   // needs the long way: set not cached / a hash collision / further Pareto vectors
      if (KS_UNLIKELY(oddm != 0)) {
        // resolve those lanes with the full probe sequence / all Pareto vectors; a set that is not cached: the driver
        const uint64_t mm = oddm;
        uint64_t ok2 = 0, missm = 0;
        W::ballot2([&](int l) {
          if (!((mm >> l) & 1)) return 0;
          FastEnt e;
          if (fast_lookup(ent, mlv.at(l), e) < 0) return 2;
          const int32_t sz[4] = {z0[j].at(l), z1[j].at(l), z2[j].at(l), z3[j].at(l)};
          return fast_fits(pool, e, ns.req, sz) ? 1 : 0;
        }, ok2, missm);
        accm |= ok2;
        if (missm) cold_refresh = true;
      }
      accw[j] = accm;
    }
    // the claim's record and its place in the order: one lane writes
    if (W::leader()) {
      cst.put_state((uint32_t)x, ns);
#pragma unroll
      for (int j = 0; j < R; ++j) cst.put_acc((uint32_t)x, j, accw[j]);
      if (s_ <= 0) okey[a] = (uint16_t)mvn;
      else { okey[a + s_] = (uint16_t)mvn; oord[a + s_] = (uint16_t)x; }
    }
    W::sync();
    KS_SEC(ts3)   // refresh
    bi++; steps++;
    if (KS_UNLIKELY(cold_refresh)) { ev = FEV_REFRESH; ev_arg = x; break; }
    if (--budget <= 0 && pend_a < 0) { ev = FEV_CONT; break; }   // back to the fast loop (a move left pending: the next add sorts first)
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
    hs->n_steps = n_steps; hs->n_ref = n_ref; hs->ev_arg = ev_arg;
    hs->hot_cycles += W::clock() - t_in;
#ifdef KSOLVE_PHASE_TIMERS
    hs->tsec[0] += ts0; hs->tsec[1] += ts1; hs->tsec[2] += ts2; hs->tsec[3] += ts3; hs->tsec[4] += ts4; hs->tsec[5] += ts5; hs->tsec[6] += ts6; hs->tsec[7] += ts7;
#endif
  }
#undef KS_SEC
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < R; ++j) { hs->cur[j][l] = cur[j].at(l); hs->scls[j][l] = scls[j].at(l); }
    hs->nxt_cls[l] = nxt_cls.at(l); hs->bcls[l] = bcls.at(l); hs->oclaim[l] = oclaim.at(l); hs->ocnt[l] = ocnt.at(l);
  });
  W::sync();
  return ev;
}

// ---- the fast loop -------------------------------------------------------------------------------------------------------
// The same step as fast_slow_run below, for the pod whose step is the plain one — its class has a slot, a claim among the 64 at
// the class's cursor accepts it, the move of that claim is decided inside those 64 positions, every requirement set met while the
// claim is refreshed is cached — and nothing else. A step is a TRANSACTION: every test comes before the first write, and the first
// thing that is not plain (the last entry before a block where the cancel flag is polled, a pending move, a class without a slot,
// no acceptor in the window, pdqsort's other paths, a requirement set that is not cached) leaves the loop with the state as it
// was: the driver runs fast_slow_run for that one pod and comes back. Why two functions: with every rare path inside one loop the
// compiler merged ~30 loop-carried values behind each of them — 274 instructions per pod of which 140 scalar moves, selects and
// branches (SQ counters, profiles/round5); this loop has ONE path and one exit.
template <class W, int GS, int R, bool HP = false>
KS_COLD int fast_hot_run(FastHotCtx<GS, R> cx) {
  typedef typename FastMem<GS, R>::o16 o16;
  KS_LDS FastHot* const hs = fast_uniform(cx.hs);
  // (nothing to do here: a pending move / a new claim to place, a step limit — tests —, a stopped solve)
  // (... or no claim yet: the select step reads the order's entries unconditionally, clamped to the last one)
  if (fast_uniform(hs->pend_a) >= 0 || fast_uniform(hs->max_steps) >= 0 || fast_uniform(hs->status) != 0 || fast_uniform(hs->n) <= 0) return FEV_SLOW;
  const o16 okey = fast_uniform(cx.okey), oord = fast_uniform(cx.oord);
  const typename FastMem<GS, R>::States cst = fast_uniform(cx.cst);
  KS_LDS FastEnt* const ent = fast_uniform(cx.ent);
  KS_LDS int32_t* const pool = fast_uniform(cx.pool);
  KS_LDS FastSlot* const aslot = fast_uniform(cx.aslot);
  const int np = fast_uniform(hs->np), n = fast_uniform(hs->n);
  const KS_GLOBAL uint32_t* const gqcls = (const KS_GLOBAL uint32_t*)fast_uniform(hs->q_class);
  const bool polled = fast_uniform(hs->cancel) != nullptr;
  KS_GLOBAL uint32_t* const gqclaim = (KS_GLOBAL uint32_t*)fast_uniform(hs->q_claim);
  KS_GLOBAL uint32_t* const gqcnt = (KS_GLOBAL uint32_t*)fast_uniform(hs->q_cnt);
  int base = fast_uniform(hs->base), bi = fast_uniform(hs->bi), bn = fast_uniform(hs->bn), steps = fast_uniform(hs->steps);
  unsigned long long n_ref = W::uniform(hs->n_ref);
  const int bi_in = bi, base_in = base;
  LaneVar<uint32_t> cur[R], scls[R], tok[R], nxt_cls, bcls, oclaim, ocnt;
  LaneVar<uint64_t> cvm[R], dm[R], gd[R];
  LaneVar<int32_t> z0[R], z1[R], z2[R], z3[R];
  W::each([&](int l) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      cur[j].at(l) = hs->cur[j][l]; scls[j].at(l) = hs->scls[j][l];
      if constexpr (!HP) {   // (two wavefronts: the classes' records are the refresher's registers)
        const FastSlot s = lds_get(&aslot[j * 64 + l]);
        cvm[j].at(l) = s.cvmask; dm[j].at(l) = s.dmask; gd[j].at(l) = (s.dmask << 1) & ~s.dmask;
        z0[j].at(l) = s.size[0]; z1[j].at(l) = s.size[1]; z2[j].at(l) = s.size[2]; z3[j].at(l) = s.size[3];
        tok[j].at(l) = scls[j].at(l) == kFastFree ? 0u : s.tmplok;
      }
    }
    nxt_cls.at(l) = hs->nxt_cls[l]; bcls.at(l) = hs->bcls[l]; oclaim.at(l) = hs->oclaim[l]; ocnt.at(l) = hs->ocnt[l];
  });
  // ---- two wavefronts: the requests in flight at the refresher (FastMail) ----
  KS_LDS FastMail* const mail = &hs->mail;
  uint32_t seq = 0;            // the last request posted
  int pend1 = -1, pend2 = -1;  // the claims of requests seq and seq - 1 (until this wavefront waited for them)
  bool dead = false;           // the refresher does not answer (never expected: the solve is handed to the general engine)
  unsigned long long hw0 = 0, hw1 = 0, hw2 = 0, hw3 = 0, t_post = 0;   // profiling builds: waits on a claim in flight (count, cycles), on a ring slot (count), request -> seen done (cycles, at those waits)
  if constexpr (HP) {
    seq = (uint32_t)fast_uniform((int)mail->done) >> 1;   // (nothing is in flight between two runs of this loop)
    const uint32_t g = (uint32_t)fast_uniform((int)mail->gen) + 1u;
    if (W::leader()) mail->gen = g;                     // the class slots may have changed; a reported miss has been taken
    W::order();
  }
  // done counter >= s, as the refresher last reported (low bit: a miss)
  auto done_word = [&]() -> uint32_t { return (uint32_t)fast_uniform((int)mail_load(&mail->done)); };
  // wait until request s is served (emulation: serve it here — the laziest refresher there can be)
  auto wait_for = [&](uint32_t s_) -> uint32_t {
    uint32_t dw = done_word();
#if KS_DEVICE
    uint32_t spins = 0;
    while ((int32_t)((dw >> 1) - s_) < 0) { if (++spins > (1u << 22)) { dead = true; break; } dw = done_word(); }
#else
    if ((int32_t)((dw >> 1) - s_) < 0) {
      FastHelperK<R>& k = fast_emu_helper_k<R>();
      uint32_t d = dw >> 1;
      while (d != s_) { ++d; fast_helper_serve<W, GS, R>(k, cx, d, mail->word[d & 1u]); }
      dw = done_word();
    }
#endif
    W::order();
    return dw;
  };
  // the refresher's done word as it stood BEFORE the records of the pod's claims were read (the LDS serves a wavefront's reads in
  // order): a request it shows done had its acceptance word written when the gather read it
  LaneVar<uint32_t> dwl;
  auto read_done = [&]() { if constexpr (HP) { W::order(); W::each([&](int l) { dwl.at(l) = mail->done; }); } };
  // pdqsort's other paths: with 12 < n < 50 every re-sort that has something to move; with n >= 50 a move from one of choosePivot's
  // nine sampled positions (fast_sampled, as three starts); with n <= 12 none (the stable insertion sort)
  const bool inexact_b = n > 12 && n < 50;
  const uint32_t e1 = n >= 50 ? (uint32_t)(n >> 2) - 1u : 0x7FFFFFF0u, e2 = n >= 50 ? 2u * (uint32_t)(n >> 2) - 1u : 0x7FFFFFF0u, e3 = n >= 50 ? 3u * (uint32_t)(n >> 2) - 1u : 0x7FFFFFF0u;
  const int nm1 = n - 1;
  // Software pipeline over the pods of a block: a step has three dependent LDS round trips (the order's entries -> the claims'
  // records -> the requirement-set cache), and the FIRST one of the next pod does not depend on the LAST one of this pod — only
  // on this pod's writes to the order and on the cursors, both done before the cache read is waited for. So a step issues, in
  // this order: its order writes | the NEXT pod's order reads | its cache reads | its record write | the NEXT pod's record reads.
  // The LDS executes a wavefront's accesses in order, so those record reads see the write in front of them.
  unsigned long long n_ext = 0;   // four-window steps of scans beyond a cursor's window
  int rf = -1;              // a claim whose refresh met a requirement set that is not cached: the driver computes its words
  LaneVar<uint32_t> xv, kv; // the order's entries of the pod about to be placed: claim id and pod count at positions rc0 + lane
  uint32_t clsw = 0, rc0 = 0;
  uint64_t slbitA = 0;   // the bit of the class's slot in an acceptance word — none when the class has no slot: then no claim "accepts", the select comes back empty and the step is not plain (no flag of its own to carry around the loop)
  int row = 0, sl = 0;
  // the order's 64 entries at positions rr .. rr+63 (n >= 1 here)
  auto order_reads = [&](uint32_t rr) {
    W::each([&](int l) {
      const int p = (int)rr + l;
      const uint32_t pc = (uint32_t)p < (uint32_t)nm1 ? (uint32_t)p : (uint32_t)nm1;   // clamped (cursors are positions: p >= 0, and n >= 1 here): no lane is switched off for the reads (a cursor at the end of the order: no lane is valid)
      const uint32_t kk = okey[pc];
      xv.at(l) = oord[pc]; kv.at(l) = kk;   // (lanes past the order's end repeat its last entry: the window's validity mask, one scalar value per step, takes them out of the select and of the move)
    });
  };
  // the records of the claims the order read brought (state and acceptance word of the class's row), one lane each
  LaneVar<uint64_t> mvv, awv;
  LaneVar<int32_t> q0, q1, q2, q3;
  auto gather = [&]() {
    const int rws = row;
    W::each([&](int l) {
      const FastRec<R> st = cst.rec(xv.at(l));
      uint64_t aw = st.acc[0];
#pragma unroll
      for (int j = 1; j < R; ++j) aw = rws == j ? st.acc[j] : aw;
      awv.at(l) = aw; mvv.at(l) = st.vmask;
      q0.at(l) = st.req[0]; q1.at(l) = st.req[1]; q2.at(l) = st.req[2]; q3.at(l) = st.req[3];
    });
  };
  // stage A of entry i: its class slot, the class's cursor, the order's 64 entries there
  auto stage_a = [&](int i) {
    clsw = bcls.bcast(i & 63);
    const uint32_t kcls = clsw & ~kFastLastBit;
    row = 0;
    if constexpr (R == 1) {
      const uint64_t m0 = W::ballot([&](int l) { return scls[0].at(l) == kcls; });
      slbitA = m0 & (0ull - m0);
      sl = ctz64(m0 | (1ull << 63));
    } else {
      uint64_t mf = 0;
#pragma unroll
      for (int j = R - 1; j >= 0; --j) {
        const uint64_t mj = W::ballot([&](int l) { return scls[j].at(l) == kcls; });
        row = mj != 0 ? j : row; mf = mj != 0 ? mj : mf;     // (a class sits in one slot)
      }
      slbitA = mf & (0ull - mf);
      sl = ctz64(mf | (1ull << 63));
    }
    rc0 = cur[0].bcast(sl);
#pragma unroll
    for (int j = 1; j < R; ++j) { const uint32_t cj = cur[j].bcast(sl); rc0 = row == j ? cj : rc0; }
    order_reads(rc0);
  };
  for (;;) {   // blocks of the queue
    // entries of this block the loop may place: not the queue's last one, nor the last one before a block at which the cancel flag
    // is polled (a Solve() that ends there reports the order of the last sort the reference would have run: their move stays undone)
    const int bf = bn - ((base + bn >= np || (polled && ((base + 64) & 1023) == 0)) ? 1 : 0);
    if (bi < bf) { stage_a(bi); read_done(); gather(); }
    int bfx = bf;   // the loop's end: bf, or right behind the step whose refresh met a requirement set that is not cached (rf) — one compare per step for both
    while (bi < bfx) {
      // Everything up to the first write is ONE basic block: whatever is not plain sets a bit of `bad` and the step goes on with
      // harmless values (lane 0, position 0), so that no branch stands between the loads and the compiler issues them together —
      // with a `break` behind the acceptor test it had sunk the count, state and class reads below it: five dependent LDS round
      // trips per pod instead of three.
      uint64_t bad = 0;   // (booleans of the step are 64-bit masks in scalar registers: what the compiler makes of a uniform condition anyway)
      const int slot = row * 64 + sl;
      const FastSlot cs = lds_get(&aslot[slot]);
      // ---- select: the claims at the 64 positions, one lane each: the whole record (state and acceptance words) ----
      const uint64_t slbit = slbitA;
      const int rws = row;
      uint64_t vm = 0;                                      // the positions of the window that exist
      auto select = [&]() {
        const int rem = n - (int)rc0;                       // (a cursor at the end of the order: none)
        vm = rem >= 64 ? ~0ull : ((1ull << (rem < 0 ? 0 : rem)) - 1ull);
        return W::ballot([&](int l) { return (awv.at(l) & slbit) != 0; }) & vm;
      };
      uint64_t okm = select();
      if constexpr (HP) {
        // The select ran on the words as the gather found them: the claims of the last two requests may still show classes they no
        // longer accept — unless the done word read in front of the gather says their request was served. If the first acceptor is
        // such a claim: wait for the refresher, the two claims' words in the same round trip as its done word, the select again.
        const int x0 = (int)xv.bcast(ctz64(okm | (1ull << 63)));
        if (KS_UNLIKELY((okm != 0) & ((x0 == pend1) | (x0 == pend2)))) {
          const uint32_t dn = (uint32_t)fast_uniform((int)dwl.bcast(0)) >> 1;
          const bool stale1 = pend1 >= 0 && (int32_t)(dn - seq) < 0, stale2 = pend2 >= 0 && (int32_t)(dn - (seq - 1u)) < 0;
          if ((x0 == pend1 && stale1) || (x0 == pend2 && stale2)) {
#ifdef KSOLVE_PHASE_TIMERS
            const unsigned long long tw0 = W::clock();
#endif
            const uint32_t c1 = (uint32_t)(pend1 < 0 ? 0 : pend1), c2 = (uint32_t)(pend2 < 0 ? 0 : pend2);
            uint64_t a1 = 0, a2 = 0;
            uint32_t dw2 = 0;
#if KS_DEVICE
            for (uint32_t spins = 0;; ++spins) {
              const uint32_t dv = mail_load(&mail->done);
              W::order();
              const uint64_t v1 = cst.acc(c1, 0), v2 = cst.acc(c2, 0);     // (behind the done word: served means written)
              dw2 = (uint32_t)fast_uniform((int)dv); a1 = W::uniform(v1); a2 = W::uniform(v2);
              if ((int32_t)((dw2 >> 1) - seq) >= 0) break;
              if (spins > (1u << 22)) { dead = true; break; }
              W::order();
            }
#else
            dw2 = wait_for(seq); a1 = cst.acc(c1, 0); a2 = cst.acc(c2, 0);
#endif
#ifdef KSOLVE_PHASE_TIMERS
            { const unsigned long long tw1 = W::clock(); hw0++; hw1 += tw1 - tw0; hw3 += tw1 - t_post; }
#endif
            bad |= (((dw2 & 1u) != 0) | dead) ? ~0ull : 0ull;              // a requirement set that is not cached: out, nothing written
            const int p1 = pend1, p2 = pend2;
            W::each([&](int l) { const int xl = (int)xv.at(l); awv.at(l) = xl == p1 ? a1 : xl == p2 ? a2 : awv.at(l); });
            okm = select();
            pend1 = -1; pend2 = -1;                                 // (every request is served)
          } else {
            pend1 = stale1 ? pend1 : -1; pend2 = stale2 ? pend2 : -1;   // (served before the gather read their words)
          }
        }
      }
      if (KS_UNLIKELY(okm == 0)) if (slbit != 0 && (int)rc0 + 64 < n) {   // (the common case pays one compare: the other two only when no claim of the window accepts)
        // The class's next acceptor is not among the 64 claims at its cursor (the claim it was filling is full): the rest of the
        // order, four windows per step, acceptance words only, all eight reads of a step in flight before the first is used;
        // then the select step once more, at the window that holds it.
        if constexpr (HP) { if (pend1 >= 0 || pend2 >= 0) { const uint32_t dw2 = wait_for(seq); bad |= (dw2 & 1u) != 0 ? ~0ull : 0ull; pend1 = -1; pend2 = -1; } }   // (exact words for the scan)
        int r = (int)rc0 + 64, found = -1;
        while (r < n && found < 0) {
          uint64_t m4[4];
          LaneVar<uint32_t> x4[4];
          W::each([&](int l) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int p = r + 64 * j + l; x4[j].at(l) = oord[p < nm1 ? p : nm1]; }
          });
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int rj = r + 64 * j;
            m4[j] = W::ballot([&](int l) { return ((cst.acc(x4[j].at(l), rws) & slbit) != 0) & (rj + l < n); });
          }
          n_ext++;
#pragma unroll
          for (int j = 3; j >= 0; --j) found = m4[j] != 0 ? r + 64 * j : found;
          r += 256;
        }
        if (found >= 0) {     // (none: no in-flight claim accepts the pod — addToNewNodeClaim, through fast_slow_run)
          rc0 = (uint32_t)found;
          order_reads(rc0);
          gather();
          okm = select();
        }
      }
      const int first_ok = ctz64(okm | (1ull << 63));   // (no acceptor: lane 63, and the reach test below says "not plain")
      const int a = (int)rc0 + first_ok;
      const int x = (int)xv.bcast(first_ok);
      const uint32_t cnt = kv.bcast(first_ok);
      bad |= cnt >= 65534u ? ~0ull : 0ull;
      // ---- NodeClaim.Add (nodeclaim.go:247-263): the claim's new state (every lane for the claim it read; lane first_ok's is the one) ----
      FastClaim ns;
      LaneVar<uint64_t> nmv; LaneVar<int32_t> n0v, n1v, n2v, n3v;
      W::each([&](int l) {
        nmv.at(l) = mvv.at(l) & cs.cvmask;
        n0v.at(l) = q0.at(l) + cs.size[0]; n1v.at(l) = q1.at(l) + cs.size[1]; n2v.at(l) = q2.at(l) + cs.size[2]; n3v.at(l) = q3.at(l) + cs.size[3];
      });
      ns.vmask = nmv.bcast(first_ok);
      ns.req[0] = n0v.bcast(first_ok); ns.req[1] = n1v.bcast(first_ok); ns.req[2] = n2v.bcast(first_ok); ns.req[3] = n3v.bcast(first_ok);
      // ---- the move of the next add's sort.Slice (scheduler.go:598), decided from the counts the order read brought ----
      const uint32_t mvn = cnt + 1;
      const uint64_t lessm = W::ballot([&](int l) { return kv.at(l) < mvn; }) & vm;   // (the lanes up to first_ok are shifted out)
      const uint64_t tsh = (lessm >> 1) >> first_ok;                             // (two shifts: first_ok may be 63)
      const int sm = ctz64(~tsh);                                                // < 64: the top bit of tsh is clear
      bad |= first_ok + 1 + sm >= 64 ? ~0ull : 0ull;   // the move reaches the window's last lane or beyond (the pending path), or no claim of the window accepts (first_ok = 63) — a window over the order's tail whose claim lands on lane 63 goes the long way too: rare, exact
      {
        const uint32_t d1 = (uint32_t)a - e1, d2 = (uint32_t)a - e2, d3 = (uint32_t)a - e3;
        const uint32_t dmin = fast_umin_s(fast_umin_s(d1, d2), d3);
        bad |= ((sm != 0) & (inexact_b | (dmin <= 2u))) ? ~0ull : 0ull;   // pdqsort's other paths
      }
      if (KS_UNLIKELY(bad != 0)) { bfx = bi; continue; }   // (out through the loop's own test: one exit)
      // ---- nothing has been written so far; from here on the step is the plain one ----
      if constexpr (HP) {
        static_assert(!HP || R == 1, "the two-wavefront loop: one row of class slots");
        // The claim's new state goes to its record and the claim, as request seq + 1, to the refresher — FIRST, so that the refresher
        // works while this wavefront moves the claim in the order, steps the cursors and reads the next pod's claims. The request
        // takes the ring slot of request seq - 1, which must be done (it is, unless the refresher is the slower of the two).
        {
          const uint32_t dw = (uint32_t)fast_uniform((int)dwl.bcast(0));
          if (KS_UNLIKELY(((int32_t)((dw >> 1) - (seq - 1u)) < 0) | ((dw & 1u) != 0))) {
            const uint32_t dw2 = wait_for(seq - 1u);
#ifdef KSOLVE_PHASE_TIMERS
            hw2++;
#endif
            if ((dw2 & 1u) | (uint32_t)dead) break;             // a requirement set that is not cached: out, nothing written
          }
        }
        ++seq;
        const int fo = first_ok; const uint32_t sq = seq;
        W::each([&](int l) {
          if (l == fo) {
            FastClaim mine;
            mine.vmask = nmv.at(l); mine.req[0] = n0v.at(l); mine.req[1] = n1v.at(l); mine.req[2] = n2v.at(l); mine.req[3] = n3v.at(l);
            cst.put_state(xv.at(l), mine);
            W::order();
            mail_store(&mail->word[sq & 1u], (sq << 16) | xv.at(l));
          }
        });
        pend2 = pend1; pend1 = x;
#ifdef KSOLVE_PHASE_TIMERS
        t_post = W::clock();
#endif
#if !KS_DEVICE
        if (fast_emu_helper_eager()) wait_for(seq);   // (emulation: the refresher at its fastest; by default at its laziest)
#endif
        // the order: lanes first_ok+1 .. first_ok+sm step one position to the left, the claim lands behind them with its new count
        {
          const int rb = (int)rc0, smv = sm;
          W::each([&](int l) {
            if (l >= fo && l <= fo + smv) {
              const bool me = l == fo;
              const int dst = me ? rb + l + smv : rb + l - 1;
              okey[dst] = (uint16_t)(me ? kv.at(l) + 1u : kv.at(l)); oord[dst] = (uint16_t)xv.at(l);
            }
          });
        }
        W::order();
        // The next pod's stage A on the cursors AS THEY WILL BE — its class's cursor, read before the update below and corrected in
        // scalar registers (this class's cursor comes to a; a cursor in (a, a+sm] steps left) — so that the order's entries are on
        // their way while the vector registers are brought up to date. (The slot match runs on the slots as they were: a class that
        // leaves with this pod is not the next pod's.)
        const uint32_t ua1 = (uint32_t)a + 1u, su = (uint32_t)sm;
        const uint32_t lastm = (uint32_t)((int32_t)clsw >> 31);   // all ones on the class's last entry (kFastLastBit is the sign bit)
        const int slot_now = slot;
        {
          clsw = bcls.bcast((bi + 1) & 63);
          const uint32_t kcls = clsw & ~kFastLastBit;
          const uint64_t m0 = W::ballot([&](int l) { return scls[0].at(l) == kcls; });
          slbitA = m0 & (0ull - m0);
          sl = ctz64(m0 | (1ull << 63));
          const uint32_t rr = cur[0].bcast(sl);
          rc0 = sl == slot_now ? (uint32_t)a : rr - (uint32_t)((rr - ua1) < su);
          order_reads(rc0);
        }
        // cursors in (a, a+sm] step left; the pod's result; the class's cursor comes to a; its slot is free after its last entry
        {
          const int bb = bi;
          W::each([&](int l) {
            const bool mine = l == bb;
            oclaim.at(l) = mine ? (uint32_t)x : oclaim.at(l); ocnt.at(l) = mine ? cnt : ocnt.at(l);
            const bool me = l == slot_now;
            const uint32_t rr = cur[0].at(l);
            cur[0].at(l) = me ? (uint32_t)a : rr - (uint32_t)((rr - ua1) < su);
            scls[0].at(l) |= me ? lastm : 0u;
          });
        }
        n_ref += (unsigned long long)ua1;
        read_done();
        gather();
        bi++; steps++;
        continue;
      }
      {
        // The order: lanes first_ok+1 .. first_ok+sm (the claims with a smaller count) step one position to the left, the claim
        // lands behind them with its new count (lane first_ok writes that entry)
        const int rb = (int)rc0, fo = first_ok, smv = sm;
        W::each([&](int l) {
          if (l >= fo && l <= fo + smv) {
            const bool me = l == fo;
            const int dst = me ? rb + l + smv : rb + l - 1;
            okey[dst] = (uint16_t)(me ? kv.at(l) + 1u : kv.at(l)); oord[dst] = (uint16_t)xv.at(l);
          }
        });
      }
      {
        // cursors in (a, a+sm] step left; the pod's result; the class's cursor comes to a; its slot is free after its last entry
        const uint32_t ua1 = (uint32_t)a + 1u, su = (uint32_t)sm;
        const uint32_t lastm = (uint32_t)((int32_t)clsw >> 31);   // all ones on the class's last entry (kFastLastBit is the sign bit)
        const int bb = bi;
        W::each([&](int l) {
          const bool mine = l == bb;
          oclaim.at(l) = mine ? (uint32_t)x : oclaim.at(l); ocnt.at(l) = mine ? cnt : ocnt.at(l);
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const bool me = j * 64 + l == slot;
            const uint32_t fm = me ? lastm : 0u;     // (branch-free: a branch taken once per class measured slower than these three instructions per pod, pass N)
            const uint32_t rr = cur[j].at(l);
            cur[j].at(l) = me ? (uint32_t)a : rr - (uint32_t)((rr - ua1) < su);
            scls[j].at(l) |= fm; tok[j].at(l) &= ~fm;
          }
        });
      }
      n_ref += (unsigned long long)((uint32_t)a + 1u);
      // ---- refresh: CanAdd (nodeclaim.go:124-242) of the claim as it stands now, for the classes of all slots (lane = slot):
      // the next pod's order reads go out first, the cache reads behind them (both in flight together), then the predicates ----
      const uint32_t tbit = 1u << (uint32_t)(ns.vmask >> 56);
      LaneVar<uint64_t> mlv[R], evm[R];
      LaneVar<int32_t> c0[R], c1[R], c2[R], c3[R];
      // several rows of class slots: the two predicates that need no cache entry first (the class tolerates the claim's template; on
      // every key the class selects on the claim keeps a value) — a row in which no class passes them has no bit to compute: no cache
      // read, no fit tests (new_slot keeps classes that exclude each other in different rows)
      uint64_t rowm[R];
      if constexpr (R > 1) {
        W::each([&](int l) {
#pragma unroll
          for (int j = 0; j < R; ++j) mlv[j].at(l) = ns.vmask & cvm[j].at(l);
        });
#pragma unroll
        for (int j = 0; j < R; ++j)
          rowm[j] = W::ballot([&](int l) { return (tok[j].at(l) & tbit) != 0; }) &
                    W::ballot([&](int l) { return (((mlv[j].at(l) & dm[j].at(l)) + dm[j].at(l)) & gd[j].at(l)) == gd[j].at(l); });
      } else rowm[0] = ~0ull;
      W::each([&](int l) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (R > 1 && rowm[j] == 0) continue;
          const uint64_t m = ns.vmask & cvm[j].at(l);
          const FastEnt e = ent_live(&ent[fast_hash(m)]);
          mlv[j].at(l) = m; evm[j].at(l) = e.vmask;
          c0[j].at(l) = e.cap[0]; c1[j].at(l) = e.cap[1]; c2[j].at(l) = e.cap[2]; c3[j].at(l) = e.cap[3];
        }
      });
      const int fo_ = first_ok;
      LaneVar<uint32_t> xw;   // (the claim ids of this step: the record write below still needs lane first_ok's)
      W::each([&](int l) { xw.at(l) = xv.at(l); });
      if constexpr (FastMem<GS, R>::kOrderHbm) W::hbm_sync(); else W::order();   // (an order in HBM: the stores are waited for, as everywhere in this engine; in LDS nothing is)
      W::sched_fence();
      stage_a(bi + 1);      // entry bi+1 of the block (entry 64 of a full block does not exist: its values are never used)
      W::sched_fence();
      uint64_t accw[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (R > 1 && rowm[j] == 0) { accw[j] = 0; continue; }
        // every predicate is one compare whose result is the 64-lane mask; the masks are combined in scalar registers
        const uint64_t tokm = R > 1 ? rowm[j] : W::ballot([&](int l) { return (tok[j].at(l) & tbit) != 0; });
        const uint64_t fldm = R > 1 ? ~0ull : W::ballot([&](int l) { return (((mlv[j].at(l) & dm[j].at(l)) + dm[j].at(l)) & gd[j].at(l)) == gd[j].at(l); });   // (several rows: inside rowm already)
        const uint64_t simm = W::ballot([&](int l) { return evm[j].at(l) == mlv[j].at(l); });
        const uint64_t f0 = W::ballot([&](int l) { return z0[j].at(l) <= c0[j].at(l) - ns.req[0]; });
        const uint64_t f1 = W::ballot([&](int l) { return z1[j].at(l) <= c1[j].at(l) - ns.req[1]; });
        const uint64_t f2 = W::ballot([&](int l) { return z2[j].at(l) <= c2[j].at(l) - ns.req[2]; });
        const uint64_t f3 = W::ballot([&](int l) { return z3[j].at(l) <= c3[j].at(l) - ns.req[3]; });
        const uint64_t basem = tokm & fldm, fitm = f0 & f1 & f2 & f3;
        uint64_t accm = basem & simm & fitm;
        const uint64_t oddm = basem & ~simm;   // set not at its first probe / not cached / an entry with further Pareto vectors (kFastExtBit)
        if (KS_UNLIKELY(oddm != 0)) {
          uint64_t ok2 = 0, missm = 0;
          W::ballot2([&](int l) {
            if (!((oddm >> l) & 1)) return 0;
            FastEnt e;
            if (fast_lookup(ent, mlv[j].at(l), e) < 0) return 2;
            const int32_t sz[4] = {z0[j].at(l), z1[j].at(l), z2[j].at(l), z3[j].at(l)};
            return fast_fits(pool, e, ns.req, sz) ? 1 : 0;
          }, ok2, missm);
          accm |= ok2;
          if (missm) { rf = x; bfx = bi + 1; }
        }
        accw[j] = accm;
      }
      // the claim's record: its state (lane first_ok computed it for the claim it read) and the acceptance words
      W::each([&](int l) {
        if (l == fo_) {
          FastRec<R> mine;
          mine.vmask = nmv.at(l); mine.req[0] = n0v.at(l); mine.req[1] = n1v.at(l); mine.req[2] = n2v.at(l); mine.req[3] = n3v.at(l);
#pragma unroll
          for (int j = 0; j < R; ++j) mine.acc[j] = accw[j];
          cst.put_rec(xw.at(l), mine);
        }
      });
      if constexpr (FastMem<GS, R>::kStateHbm) W::hbm_sync(); else W::order();   // (LDS: no wait — it executes a wavefront's accesses in order, the reads below see these writes)
      gather();     // the next pod's claims (behind this pod's record write): in flight while the loop comes around
      bi++; steps++;
    }
    if (bi < bf || bf < bn || rf >= 0) break;          // a pod the loop does not place / the block's last entry is not the loop's
    // ---- the block is done: its results, the next block ----
    if (bn > 0) {
      const int dn = bn, b0 = base;
      W::each([&](int l) { if (l < dn) { gqclaim[b0 + l] = oclaim.at(l); gqcnt[b0 + l] = ocnt.at(l); } });
    }
    const int nbase = bn > 0 ? base + 64 : base;
    if (nbase >= np || (polled && (nbase & 1023) == 0)) break;   // the end of the queue, a block at which the cancel flag is polled: fast_slow_run
    base = nbase;
    bn = np - base < 64 ? np - base : 64;
    bi = 0;
    const int nb = base + 64;
    W::each([&](int l) { bcls.at(l) = nxt_cls.at(l); });
    W::each([&](int l) { if (nb + l < np) nxt_cls.at(l) = gqcls[nb + l]; });
  }
  // ---- two wavefronts: nothing stays in flight between two runs of this loop; a miss the refresher reported is the driver's ----
  int rf2 = -1;
  if constexpr (HP) {
    const uint32_t dw = wait_for(seq);
    if (KS_UNLIKELY((dw & 1u) != 0)) {
      rf = fast_uniform((int)mail->miss_x[0]); rf2 = fast_uniform((int)mail->miss_x[1]);
      if (rf < 0) { rf = rf2; rf2 = -1; }
      W::order();
      if (W::leader()) { mail->miss_x[0] = -1; mail->miss_x[1] = -1; mail_store(&mail->done, seq << 1); }
      W::order();
    }
    if (KS_UNLIKELY(dead)) return FEV_DEAD;
#ifdef KSOLVE_PHASE_TIMERS
    if (W::leader()) { hs->hw[0] += hw0; hs->hw[1] += hw1; hs->hw[2] += hw2; hs->hw[3] += hw3; }
#endif
  }
  // ---- state out (only what this function changes) ----
  if (bi != bi_in || base != base_in) {
    if (W::leader()) { hs->base = base; hs->bi = bi; hs->bn = bn; hs->steps = steps; hs->n_ref = n_ref; hs->rf_x = rf; hs->rf_x2 = rf2; hs->n_steps += 4 * n_ext; }
    W::each([&](int l) {
#pragma unroll
      for (int j = 0; j < R; ++j) { hs->cur[j][l] = cur[j].at(l); hs->scls[j][l] = scls[j].at(l); }
      hs->nxt_cls[l] = nxt_cls.at(l); hs->bcls[l] = bcls.at(l); hs->oclaim[l] = oclaim.at(l); hs->ocnt[l] = ocnt.at(l);
    });
    W::sync();
  }
  return FEV_SLOW;
}

// The driver: runs the loop, handles its events through FastCold.
template <class W, int GS = 0, int R = 1, bool HP = false>
struct FastEngine {
  FastCold<W, GS, R> cold;
  KS_LDS FastHot* hs;
  KS_DEV FastEngine(const ProblemView* p, const Workspace* s, const FastWork* f, char* lds) { cold.init(p, s, f, lds); hs = cold.hs; }

  KS_DEV void solve() {
    KS_LDS FastHot* const h = fast_uniform(hs);
    W::each([&](int l) { for (int j = 0; j < kFastRows; ++j) { h->cur[j][l] = 0; h->scls[j][l] = kFastFree; } });
    W::sync();
    {
      const int why = (int)W::uniform((uint64_t)(uint32_t)cold.setup());
      if (why) { cold.bail_code = why; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
    }
    const int np = fast_uniform(cold.Pk->n_pods);
    if (W::leader()) {
      h->base = 0; h->bi = 0; h->bn = 0; h->n = 0; h->np = np; h->steps = 0; h->status = 0;
      const long long ms_ = cold.Sk->max_steps;
      h->max_steps = ms_ < 0 ? -1 : (int)(ms_ > 0x7FFFFFFF ? 0x7FFFFFFF : ms_);
      h->pend_a = -1; h->pend_x = 0; h->pend_mv = 0; h->pend_new = 0; h->ev_arg = 0; h->rf_x = -1; h->rf_x2 = -1;
      h->n_steps = 0; h->n_tests = 0; h->n_ref = 0; h->hot_cycles = 0;
      for (int i = 0; i < 8; ++i) h->tsec[i] = 0;
      for (int i = 0; i < 4; ++i) h->hw[i] = 0;
      h->q_class = cold.Fk->q_class; h->cancel = cold.Sk->cancel_flag;
      h->q_claim = cold.Fk->q_claim; h->q_cnt = cold.Fk->q_cnt;
    }
    {
      const uint32_t* qc = cold.Fk->q_class;
      W::each([&](int l) { h->nxt_cls[l] = l < np ? qc[l] : 0; h->bcls[l] = 0; h->oclaim[l] = 0; h->ocnt[l] = 0; });
    }
    W::sync();
    FastHotCtx<GS, R> cx;
    cx.okey = cold.order.key; cx.oord = cold.order.ord; cx.cst = cold.cst; cx.ent = cold.ent; cx.pool = cold.pool;
    cx.aslot = cold.aslot; cx.hs = hs;
    unsigned long long tev[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_fast = 0, n_fast = 0, t_slow = 0, n_slow = 0;   // profiling builds: shader clock inside the two loops, calls of each
    const unsigned long long t_begin = W::clock();
    const bool use_fast = fast_uniform(h->max_steps) < 0;   // (a step limit — tests — is counted by the general loop)
    for (;;) {
      const unsigned long long tf0 = W::clock();
      // (a new claim waits for its place in the order and the next pod is inside the current block — no block switch, no cancel poll
      // stands before the sort: the driver's event at once, without a turn through fast_slow_run)
      const bool place_now = use_fast && fast_uniform(h->pend_new) != 0 && fast_uniform(h->bi) < fast_uniform(h->bn) && fast_uniform(h->status) == 0;
      int ev = place_now ? (int)FEV_PLACE : use_fast ? fast_uniform(fast_hot_run<W, GS, R, HP>(cx)) : (int)FEV_SLOW;
      t_fast += W::clock() - tf0; n_fast++;
      if (HP && ev == FEV_DEAD) { cold.bail_code = 28; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }   // (the refresher wavefront does not answer)
      if (use_fast && fast_uniform(h->rf_x) >= 0) {
        const int rx = fast_uniform(h->rf_x), rx2 = HP ? fast_uniform(h->rf_x2) : -1;
        if (W::leader()) { h->rf_x = -1; h->rf_x2 = -1; }
        W::sync();
        if (fast_uniform(cold.refresh_claim(rx)) < 0) { cold.bail_code = 21; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
        if (rx2 >= 0 && fast_uniform(cold.refresh_claim(rx2)) < 0) { cold.bail_code = 21; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
        continue;
      }
      if (ev == FEV_SLOW) { const unsigned long long ts0 = W::clock(); ev = fast_uniform(fast_slow_run<W, GS, R>(cx, use_fast ? 1 : 0x7FFFFFFF)); t_slow += W::clock() - ts0; n_slow++; }
      if (ev == FEV_CONT) continue;
      if (ev == FEV_DONE) break;
      const unsigned long long te0 = W::clock();
      if (ev == FEV_REFRESH) {
        if (fast_uniform(cold.refresh_claim(fast_uniform(h->ev_arg))) < 0) { cold.bail_code = 21; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
      } else if (ev == FEV_SLOT) {
        if (fast_uniform(cold.new_slot(fast_uniform(h->ev_arg))) < 0) { cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
      } else if (ev == FEV_SLOWSORT || ev == FEV_PLACE) {
        const int n = fast_uniform(h->n);
        int b = -1;
        if (ev == FEV_SLOWSORT) cold.slow_sort(n, fast_uniform(h->ev_arg), 0);
        else b = fast_uniform(cold.place_new_claim(n));
        if (b == -1) {
          // pdqsort permuted positions lo..hi: cursors inside fall back to lo
          const int lo = fast_uniform(cold.lo_), hi = fast_uniform(cold.hi_);
          if (hi >= lo) W::each([&](int l) { for (int j = 0; j < R; ++j) { const uint32_t r = h->cur[j][l]; if (r > (uint32_t)lo && r <= (uint32_t)hi) h->cur[j][l] = (uint32_t)lo; } });
        } else {
          // positions [b, n-1) moved right by one; a cursor past b either steps over the new claim or — if its class is
          // accepted by it — comes back to it
          const KS_LDS uint64_t* acc = cold.Mp->acc;
          W::each([&](int l) { for (int j = 0; j < R; ++j) { const uint32_t r = h->cur[j][l]; if (r > (uint32_t)b) h->cur[j][l] = ((acc[j] >> l) & 1) ? (uint32_t)b : r + 1; } });
        }
        if (W::leader()) { h->pend_a = -1; h->pend_new = 0; }
        W::sync();
      } else if (ev == FEV_NEWCLAIM) {
        const int n = fast_uniform(h->n), bi = fast_uniform(h->bi);
        const int made = fast_uniform(cold.new_claim(fast_uniform(h->ev_arg), bi, n));
        if (!made) { cold.finish(fast_uniform(cold.bail_code) < 0 ? 1 : 3, 0, (unsigned long long)fast_uniform(h->steps), 0, 0, 0, nullptr); return; }
        if (W::leader()) { h->oclaim[bi] = (uint32_t)n; h->ocnt[bi] = 0; h->n = n + 1; h->pend_new = 1; h->pend_a = 0x7FFFFFFF; h->bi = bi + 1; }   // (pend_a: the loop tests one flag)   // claim ids are handed out in creation order
        W::sync();
      } else { cold.bail_code = 22; cold.finish(3, 0, 0, 0, 0, 0, nullptr); return; }
      if (ev >= 1 && ev <= 5) { tev[ev] += W::clock() - te0; nev[ev]++; }
    }
    // profiling builds (-DKSOLVE_PHASE_TIMERS): cycles inside the loop function, per event kind, in total; event counts
    unsigned long long tc[16] = {h->hot_cycles, tev[1], tev[2], tev[3], tev[4], tev[5], W::clock() - t_begin, nev[3] + (nev[4] << 20) + (nev[1] << 40),
                                 t_fast, n_fast, t_slow, n_slow, nev[1], nev[2], nev[3] + (nev[4] << 32), nev[5]};
    const unsigned long long steps = (unsigned long long)fast_uniform(h->steps);
    // CanAdd evaluations of the loop: after every placement the claim against the classes of all slots; select steps: one per pod and the extra ones
    cold.finish(fast_uniform(h->status), fast_uniform(h->n), steps, h->n_steps + steps, steps * (unsigned long long)(64 * R), h->n_ref, tc);
  }
};

// ksolve_fast_queue   — one thread per queue entry, before the loop: its class, "not placed", and the class's first / last entry
// ksolve_fast_overlap — one thread per class: how many classes are live at the class's first entry (the most over all classes =
//                       the slots the loop needs so that no class ever loses its slot); marks nothing
// ksolve_fast_mark    — one thread per queue entry: kFastLastBit on the last entry of its class
// ksolve_fast_scatter — one thread per queue entry, after the loop: the entry's result under its pod index (Results.pod_assignment / pod_slot)
struct FastQueueArgs {
  const uint32_t* sorted; const uint32_t* row_class;
  uint32_t* q_class; uint32_t* q_claim; uint32_t* q_cnt;
  int32_t* assign; uint32_t* slot;
  uint32_t* cls_first; uint32_t* cls_last; uint32_t* max_active;
};
KS_DEV void fast_queue_body(int i, const FastQueueArgs& a) {
  const uint32_t c = a.row_class[a.sorted[i]];
  a.q_class[i] = c; a.q_claim[i] = 0xFFFFFFFFu;
  // (a plain read first: most entries lie inside what other threads have published already)
  if (a.cls_first[c] > (uint32_t)i) atomic_min_u32(&a.cls_first[c], (uint32_t)i);
  if (a.cls_last[c] < (uint32_t)i) atomic_max_u32(&a.cls_last[c], (uint32_t)i);
}
KS_DEV void fast_overlap_body(int c, int nc, const FastQueueArgs& a) {
  const uint32_t f = a.cls_first[c];
  if (f == 0xFFFFFFFFu) return;
  uint32_t live = 0;
  for (int o = 0; o < nc; ++o) live += (uint32_t)((a.cls_first[o] <= f) & (a.cls_last[o] >= f));
  if (*a.max_active < live) atomic_max_u32(a.max_active, live);
}
KS_FN void fast_mark_body(int i, const FastQueueArgs& a) {
  const uint32_t c = a.q_class[i];
  if (a.cls_last[c] == (uint32_t)i) a.q_class[i] = c | kFastLastBit;
}
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
