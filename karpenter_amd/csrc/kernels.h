// kernels.h — the data-parallel kernels around the pack engine (bodies; the __global__ wrappers are in ksolve.hip).
//
//   it_index_*   : inverts InstanceType.Requirements into per-(key,value) instance-type bitmasks, so that
//                  "which instance types intersect this requirement set" is a handful of ORs (engine.h compat_mask)
//   row_hash / row_insert / row_verify / class_number / row_class / class_gather :
//                  pod equivalence classes. Streams the per-pod SoA rows (requests, requirement masks, toleration and
//                  topology masks) once from HBM, hashes each row, and deduplicates through an open-addressing table so
//                  that requirement algebra and pruning state are kept per class, not per pod. HBM-bound.
//   sort_key_*   : queue order keys (queue.go:72-108): cpu desc, memory desc, creationTimestamp asc, uid asc
//   finalize     : per claim cheapest compatible available offering (the packing-cost estimator)
#pragma once
#include "ksp.h"
#include "nodecheck.h"
#include "go_sort.h"

namespace ks {

#if KS_DEVICE
KS_DEV uint64_t atomic_cas_u64(uint64_t* p, uint64_t expect, uint64_t v) { return (uint64_t)atomicCAS((unsigned long long*)p, (unsigned long long)expect, (unsigned long long)v); }
KS_DEV uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { return atomicMin(p, v); }
KS_DEV uint32_t atomic_max_u32(uint32_t* p, uint32_t v) { return atomicMax(p, v); }
KS_DEV uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
KS_DEV void atomic_min_i64(int64_t* p, int64_t v) { atomicMin((long long*)p, (long long)v); }
KS_DEV void atomic_or_u64(uint64_t* p, uint64_t v) { atomicOr((unsigned long long*)p, (unsigned long long)v); }
KS_DEV void atomic_or_u32(uint32_t* p, uint32_t v) { atomicOr(p, v); }
#else
inline uint64_t atomic_cas_u64(uint64_t* p, uint64_t expect, uint64_t v) { uint64_t o = *p; if (o == expect) *p = v; return o; }
inline uint32_t atomic_min_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
inline uint32_t atomic_max_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
inline uint32_t atomic_add_u32(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
inline void atomic_min_i64(int64_t* p, int64_t v) { if (v < *p) *p = v; }
inline void atomic_or_u64(uint64_t* p, uint64_t v) { *p |= v; }
inline void atomic_or_u32(uint32_t* p, uint32_t v) { *p |= v; }
#endif

// ------------------------------------------------------------------------------------------------ instance types
struct ItIndexArgs {
  Dict dict;
  int n_its, it_words, n_res;
  ReqTable it_reqs;
  const int64_t* it_alloc;
  uint64_t* kv_has;      // [req_words*64][it_words]
  uint64_t* key_undef;   // [n_keys][it_words]
  uint64_t* key_compl;
  uint64_t* key_neg;
  uint64_t* it_alloc_ok; // [it_words]
  uint32_t* error;       // bit0: an instance type does not require In [own name] on the instance-type key; bit1: bounds on an instance type
};
// one thread per instance type
KS_FN void it_index_body(int it, const ItIndexArgs& a) {
  const Dict& d = a.dict;
  ReqRef r = a.it_reqs.at(d, it);
  const int iw = a.it_words;
  const uint64_t bitv = 1ull << (it & 63);
  const int word = it >> 6;
  if (r.has_gte | r.has_lte) atomic_or_u32(a.error, 2u);
  bool ok = true;
  for (int x = 0; x < a.n_res; ++x) ok = ok && a.it_alloc[(size_t)x * a.n_its + it] >= 0;
  if (ok) atomic_or_u64(&a.it_alloc_ok[word], bitv);
  for (int k = 0; k < d.n_keys; ++k) {
    uint32_t w0 = d.key_word_off[k], w1 = d.key_word_off[k + 1];
    if (k == d.key_it) {
      // must be exactly In [own name]: engine.h relies on dictionary index == instance type index
      bool good = bit(r.defined, k) && !bit(r.complement, k);
      for (uint32_t w = w0; w < w1 && good; ++w) good = r.mask[w] == (((int)(w - w0) == word) ? bitv : 0ull);
      if (!good) atomic_or_u32(a.error, 1u);
      continue;
    }
    if (!bit(r.defined, k)) { atomic_or_u64(&a.key_undef[(size_t)k * iw + word], bitv); continue; }
    bool comp = bit(r.complement, k);
    if (comp) atomic_or_u64(&a.key_compl[(size_t)k * iw + word], bitv);
    if (op_negative(req_op(d, r, k))) atomic_or_u64(&a.key_neg[(size_t)k * iw + word], bitv);
    for (uint32_t w = w0; w < w1; ++w) {
      uint64_t has = comp ? (~r.mask[w] & d.value_valid[w]) : r.mask[w];  // Requirement.Has(value) over the dictionary
      while (has) {
        int b = ctz64(has);
        has &= has - 1;
        atomic_or_u64(&a.kv_has[((size_t)w * 64 + b) * iw + word], bitv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ pod classes
struct RowArgs {
  Dict dict;
  int n_rows, n_res;
  const int64_t* requests;  // [n_res][n_rows]
  ReqTable reqs, strict;
  const uint64_t* tolerates;
  const uint64_t* host_ports;     // [n_rows][2] host-port triples bound | matched (nullptr: no pod binds one)
  uint64_t* cls_host_ports;       // [n_classes][2]
  const uint64_t* vol;            // [n_rows] volume requirement alternatives: first | count << 32 (nullptr: no pod has any)
  uint64_t* cls_vol;              // [n_classes]
  const uint64_t* topo_owned;     // [n_rows][topo_words] or nullptr
  const uint64_t* topo_selected;
  int topo_words;
  // table
  uint64_t seed;
  uint64_t hash_keep;       // bits of the row hash that are used (all ones; narrowed only by the collision-detection test)
  uint32_t table_size;      // power of two
  uint64_t* table_hash;     // 0 = empty
  uint32_t* table_rep;      // smallest row with that hash
  uint32_t* table_class;
  uint32_t* row_slot;
  uint32_t* row_class;
  uint32_t* n_classes;
  uint32_t* collision;
  // class tables
  uint32_t* class_rep;
  int64_t* cls_requests;    // [n_classes][n_res]
  MutReqTable cls_reqs, cls_strict;
  uint64_t* cls_tolerates;
  int64_t* min_request;     // [n_res]
  uint64_t* cls_hot;        // [n_classes][k_hot_words]
  uint64_t* cls_cold;       // [n_classes][cold_words]
  uint64_t* cls_topo;       // [n_classes][2*topo_words] topology groups owned | selected
  RecLayout lay;
};
KS_FN uint64_t mix64(uint64_t h, uint64_t v) {
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 29;
  return h;
}
KS_FN uint64_t hash_reqset(const Dict& d, uint64_t h, const ReqRef& r) {
  h = mix64(h, ((uint64_t)r.defined << 32) | r.complement);
  h = mix64(h, ((uint64_t)r.has_gte << 32) | r.has_lte);
  uint32_t keys = r.defined;
  while (keys) {
    int k = __builtin_ctz(keys);
    keys &= keys - 1;
    for (uint32_t w = d.key_word_off[k]; w < d.key_word_off[k + 1]; ++w) h = mix64(h, r.mask[w]);
    if (bit(r.has_gte, k)) h = mix64(h, (uint64_t)r.gte[k]);
    if (bit(r.has_lte, k)) h = mix64(h, (uint64_t)r.lte[k]);
    if (r.minv) h = mix64(h, (uint64_t)(int64_t)r.minv[k]);
  }
  return h;
}
KS_FN bool equal_reqset(const Dict& d, const ReqRef& a, const ReqRef& b) {
  if (a.defined != b.defined || a.complement != b.complement || a.has_gte != b.has_gte || a.has_lte != b.has_lte) return false;
  uint32_t keys = a.defined;
  while (keys) {
    int k = __builtin_ctz(keys);
    keys &= keys - 1;
    for (uint32_t w = d.key_word_off[k]; w < d.key_word_off[k + 1]; ++w) if (a.mask[w] != b.mask[w]) return false;
    if (bit(a.has_gte, k) && a.gte[k] != b.gte[k]) return false;
    if (bit(a.has_lte, k) && a.lte[k] != b.lte[k]) return false;
    int32_t am = a.minv ? a.minv[k] : -1, bm = b.minv ? b.minv[k] : -1;
    if (am != bm) return false;
  }
  return true;
}
KS_FN bool rows_equal(const RowArgs& a, int x, int y) {
  for (int r = 0; r < a.n_res; ++r) if (a.requests[(size_t)r * a.n_rows + x] != a.requests[(size_t)r * a.n_rows + y]) return false;
  if (!equal_reqset(a.dict, a.reqs.at(a.dict, x), a.reqs.at(a.dict, y))) return false;
  if (!equal_reqset(a.dict, a.strict.at(a.dict, x), a.strict.at(a.dict, y))) return false;
  if (a.tolerates[x] != a.tolerates[y]) return false;
  if (a.host_ports && (a.host_ports[(size_t)x * 2] != a.host_ports[(size_t)y * 2] || a.host_ports[(size_t)x * 2 + 1] != a.host_ports[(size_t)y * 2 + 1])) return false;
  if (a.vol && a.vol[x] != a.vol[y]) return false;
  if (a.topo_owned) for (int w = 0; w < a.topo_words; ++w) {
    if (a.topo_owned[(size_t)x * a.topo_words + w] != a.topo_owned[(size_t)y * a.topo_words + w]) return false;
    if (a.topo_selected[(size_t)x * a.topo_words + w] != a.topo_selected[(size_t)y * a.topo_words + w]) return false;
  }
  return true;
}
// One pass over the pod rows (the 421 B/row stream of the classing prepass): hash the row, find its slot, keep the smallest
// row of the slot as the class representative, and check the row against a row that reached the slot earlier — equality
// is transitive, so every row of a slot equals its first row unless a 64-bit hash collision is reported (the host
// re-seeds). The table is read before it is written: with a few thousand classes for a million rows nearly every row
// finds its hash and a smaller representative already there and issues no atomic at all (a stale read only shows an
// older state — empty slot, larger representative — and falls through to the atomic).
// bit w of (m0, m1): mask word w belongs to a key the set defines
// (A branch-free form — every key's word range as uniform masks, selected by the lane's `defined` bit — measured slower on the
// MI355X, 87-89 us against 77 us for the classing kernel: rows define a few of the 13 keys, skipping the others is cheaper.)
KS_FN void word_defined_mask(const Dict& d, uint32_t defined, uint64_t& m0, uint64_t& m1) {
  m0 = 0; m1 = 0;
  for (int k = 0; k < d.n_keys; ++k) {
    if (!bit(defined, k)) continue;
    for (uint32_t w = d.key_word_off[k]; w < d.key_word_off[k + 1]; ++w) { if (w < 64) m0 |= 1ull << w; else m1 |= 1ull << (w - 64); }
  }
}
// The difference of two requirement sets whose words are near (LDS): no early exit on the mask words, every word is loaded
// whatever the words before it held, so the loads are independent and in flight together (an early exit per word makes
// 2 x req_words DEPENDENT round trips out of one comparison). Words of keys the set does not define are ignored, as in
// equal_reqset.
KS_FN uint64_t reqset_diff(const Dict& d, const ReqRef& a, const ReqRef& b) {
  uint64_t diff = (uint64_t)((a.defined ^ b.defined) | (a.complement ^ b.complement) | (a.has_gte ^ b.has_gte) | (a.has_lte ^ b.has_lte));
  // bit w of wdef: word w belongs to a key the set defines
  uint64_t wdef0, wdef1;
  word_defined_mask(d, a.defined, wdef0, wdef1);
  const int rw = d.req_words;
  int w = 0;
  for (; w + 4 <= rw; w += 4) {
    const uint64_t b0 = b.mask[w], b1 = b.mask[w + 1], b2 = b.mask[w + 2], b3 = b.mask[w + 3];
    const uint64_t a0 = a.mask[w], a1 = a.mask[w + 1], a2 = a.mask[w + 2], a3 = a.mask[w + 3];
    const uint64_t sel = w < 64 ? wdef0 >> w : wdef1 >> (w - 64);   // four-word groups never straddle bit 64
    diff |= (a0 ^ b0) & (0ull - (sel & 1)); diff |= (a1 ^ b1) & (0ull - ((sel >> 1) & 1));
    diff |= (a2 ^ b2) & (0ull - ((sel >> 2) & 1)); diff |= (a3 ^ b3) & (0ull - ((sel >> 3) & 1));
  }
  for (; w < rw; ++w) { const uint64_t sel = w < 64 ? wdef0 >> w : wdef1 >> (w - 64); diff |= (a.mask[w] ^ b.mask[w]) & (0ull - (sel & 1)); }
  if (a.minv != nullptr || b.minv != nullptr) {
    // minValues of the defined keys: every key's pair is loaded, the comparison is masked (independent loads again)
    const int nk = d.n_keys;
    for (int k = 0; k < nk; ++k) {
      const int32_t am = a.minv ? a.minv[k] : -1, bm = b.minv ? b.minv[k] : -1;
      diff |= (uint64_t)(uint32_t)(am ^ bm) & (0ull - (uint64_t)((a.defined >> k) & 1u));
    }
  }
  if (a.has_gte | a.has_lte) {
    uint32_t keys = a.defined;
    while (keys) {
      int k = __builtin_ctz(keys);
      keys &= keys - 1;
      if (bit(a.has_gte, k) && a.gte[k] != b.gte[k]) diff |= 1;
      if (bit(a.has_lte, k) && a.lte[k] != b.lte[k]) diff |= 1;
    }
  }
  return diff;
}
// q / qs: the row's two requirement sets — read from the tables (row_hash_body), or with their mask words staged in LDS by
// the wave-cooperative loader of the device kernel (ksolve.hip: ksolve_row_hash_coop), where 64 rows' masks arrive as
// fully coalesced 512-byte accesses instead of 64 lanes striding through 160-byte records.
// req_at(r) = the row's request in dimension r, tol = its toleration mask: read from the tables (row_hash_value), or already
// in registers (the wave-cooperative kernel loads them with everything else of the block in one go).
// SAME: the strict table of this launch IS the requirement table (every row's qs equals its q): the second set adds nothing to
// the hash of a row among the rows of the same launch, so it is left out — all rows of a launch are hashed by the same kernel.
// NRM: the caller's bound on n_res (8 = ksolve.h's; a launch of a problem with at most four resource dimensions says 4 and
// carries four request registers instead of eight).
template <bool SAME = false, int NRM = 8, class ReqAt>
KS_FN uint64_t row_hash_value_with(int row, const RowArgs& a, const ReqRef& q, const ReqRef& qs, ReqAt req_at, uint64_t tol) {
  uint64_t h = a.seed;
#pragma unroll
  for (int r = 0; r < NRM; ++r) if (r < a.n_res) h = mix64(h, (uint64_t)req_at(r));   // unrolled: req_at may index registers
  h = hash_reqset(a.dict, h, q);
  if (!SAME) h = hash_reqset(a.dict, h, qs);
  h = mix64(h, tol);
  if (a.host_ports) { h = mix64(h, a.host_ports[(size_t)row * 2]); h = mix64(h, a.host_ports[(size_t)row * 2 + 1]); }
  if (a.vol) h = mix64(h, a.vol[row]);
  if (a.topo_owned) for (int w = 0; w < a.topo_words; ++w) { h = mix64(h, a.topo_owned[(size_t)row * a.topo_words + w]); h = mix64(h, a.topo_selected[(size_t)row * a.topo_words + w]); }
  return h ? h : 1;
}
// hash_keep is all ones; a test narrows it to force distinct rows onto one hash and watch the verification report them
KS_FN uint64_t row_hash_kept(const RowArgs& a, uint64_t h) { h &= a.hash_keep; return h ? h : 1ull; }
KS_FN uint64_t row_hash_value(int row, const RowArgs& a, const ReqRef& q, const ReqRef& qs) {
  return row_hash_value_with(row, a, q, qs, [&](int r) { return a.requests[(size_t)r * a.n_rows + row]; }, a.tolerates[row]);
}
// The row's slot in the class table (*slot_out), and the row this one has to equal (a row that reached the slot earlier) or
// 0xFFFFFFFF when there is nothing to check. Every access here is device-wide (one address per class for all rows of the
// class): the device kernel calls it once per DISTINCT hash of a wavefront, not once per row. The representative of the
// first slot is read together with its hash — one round trip when the slot is the row's own, which it nearly always is;
// a stale value is an earlier representative of the same slot (or none yet), and the atomic below sorts that out.
KS_FN uint32_t row_table_insert(int row, const RowArgs& a, uint64_t h, uint32_t* slot_out) {
  uint32_t slot = (uint32_t)(h >> 17) & (a.table_size - 1);
  const uint32_t slot0 = slot;
  uint64_t cur = ((volatile uint64_t*)a.table_hash)[slot];
  uint32_t other = ((volatile uint32_t*)a.table_rep)[slot];
  for (;;) {
    if (cur == h) break;
    if (cur == 0ull) {
      cur = atomic_cas_u64(&a.table_hash[slot], 0ull, h);
      if (cur == 0ull) { a.table_class[slot] = atomic_add_u32(a.n_classes, 1u); break; }   // the row that opens a slot draws its class id
      if (cur == h) break;
    }
    slot = (slot + 1) & (a.table_size - 1);
    cur = ((volatile uint64_t*)a.table_hash)[slot];
  }
  *slot_out = slot;
  if (slot != slot0) other = ((volatile uint32_t*)a.table_rep)[slot];
  if (other > (uint32_t)row) other = atomic_min_u32(&a.table_rep[slot], (uint32_t)row);
  return other == (uint32_t)row ? 0xFFFFFFFFu : other;
}
// rows_equal(row, rep) for a row whose requirement sets (q, qs), requests (req_at) and toleration mask (tol) are at hand and
// a representative `rep` that is far away (HBM / L2): every access to rep's row is issued before the first one is used —
// flags, requests, minValues and the first kRowFarBatch mask words of both sets are ONE round trip, not one per field.
// Returns 0 when the rows are equal. Words of keys the set does not define are ignored, as in equal_reqset.
constexpr int kRowFarBatch = 20;   // mask words per set and batch: a 1280-value dictionary in one batch
constexpr int kRowFarKeys = 16;
// SAME: see row_hash_value_with — the strict side is the same memory, compared once. MINV = false: the launch has no minValues
// tables (the caller knows). Loads of mask words and requests past the row's end are CLAMPED, not branched around: a uniform
// branch per element costs a lone SIMD more than the load, and a word past req_words belongs to no defined key (its compare is
// masked), a request past n_res repeats request 0 on both sides.
template <bool SAME = false, bool MINV = true, int NRM = 8, class ReqAt>
KS_FN uint64_t row_diff_far(int row, const RowArgs& a, uint32_t rep, const ReqRef& q, const ReqRef& qs, ReqAt req_at, uint64_t tol) {
  const Dict& d = a.dict;
  const int rw = d.req_words, nk = d.n_keys, nr = a.n_res;
  const uint64_t* b0 = a.reqs.mask + (size_t)rep * rw;
  const uint64_t* b1 = a.strict.mask + (size_t)rep * rw;
  // ---- the loads ----
  uint64_t w0[kRowFarBatch], w1[kRowFarBatch];
#pragma unroll
  for (int i = 0; i < kRowFarBatch; ++i) { const int x = i < rw ? i : rw - 1; w0[i] = b0[x]; w1[i] = SAME ? 0ull : b1[x]; }
  const uint32_t f0 = a.reqs.defined[rep], f1 = a.reqs.complement[rep], f2 = a.reqs.has_gte ? a.reqs.has_gte[rep] : 0u, f3 = a.reqs.has_lte ? a.reqs.has_lte[rep] : 0u;
  const uint32_t s0 = SAME ? f0 : a.strict.defined[rep], s1 = SAME ? f1 : a.strict.complement[rep], s2 = SAME ? f2 : a.strict.has_gte ? a.strict.has_gte[rep] : 0u, s3 = SAME ? f3 : a.strict.has_lte ? a.strict.has_lte[rep] : 0u;
  int64_t bq[NRM];
#pragma unroll
  for (int r = 0; r < NRM; ++r) bq[r] = a.requests[(size_t)(r < nr ? r : 0) * a.n_rows + rep];
  const uint64_t btol = a.tolerates[rep];
  const int32_t* bm0 = (MINV && a.reqs.minv) ? a.reqs.minv + (size_t)rep * nk : nullptr;
  const int32_t* bm1 = (MINV && !SAME && a.strict.minv) ? a.strict.minv + (size_t)rep * nk : nullptr;
  int32_t v0[kRowFarKeys], v1[kRowFarKeys];
#pragma unroll
  for (int k = 0; k < kRowFarKeys; ++k) { v0[k] = (MINV && bm0 && k < nk) ? bm0[k] : -1; v1[k] = (MINV && !SAME && bm1 && k < nk) ? bm1[k] : -1; }
  // ---- the comparison ----
  uint64_t diff = (uint64_t)((f0 ^ q.defined) | (f1 ^ q.complement) | (f2 ^ q.has_gte) | (f3 ^ q.has_lte) |
                             (s0 ^ qs.defined) | (s1 ^ qs.complement) | (s2 ^ qs.has_gte) | (s3 ^ qs.has_lte));
  diff |= btol ^ tol;
#pragma unroll
  for (int r = 0; r < NRM; ++r) if (r < nr) diff |= (uint64_t)(bq[r] ^ req_at(r));
  uint64_t d00, d01, d10, d11;
  word_defined_mask(d, q.defined, d00, d01);
  if (SAME) { d10 = 0; d11 = 0; } else word_defined_mask(d, qs.defined, d10, d11);
#pragma unroll
  for (int i = 0; i < kRowFarBatch; ++i) {   // i >= rw: no defined key owns the word, the mask is zero (q.mask is read clamped too)
    const int x = i < rw ? i : rw - 1;
    diff |= (q.mask[x] ^ w0[i]) & (0ull - ((d00 >> i) & 1ull));
    if (!SAME) diff |= (qs.mask[x] ^ w1[i]) & (0ull - ((d10 >> i) & 1ull));
  }
  for (int w = kRowFarBatch; w < rw; w += kRowFarBatch) {   // dictionaries beyond one batch
#pragma unroll
    for (int i = 0; i < kRowFarBatch; ++i) { w0[i] = w + i < rw ? b0[w + i] : 0ull; w1[i] = (!SAME && w + i < rw) ? b1[w + i] : 0ull; }
#pragma unroll
    for (int i = 0; i < kRowFarBatch; ++i) if (w + i < rw) {
      const int x = w + i;
      diff |= (q.mask[x] ^ w0[i]) & (0ull - (((x < 64 ? d00 >> x : d01 >> (x - 64))) & 1ull));
      if (!SAME) diff |= (qs.mask[x] ^ w1[i]) & (0ull - (((x < 64 ? d10 >> x : d11 >> (x - 64))) & 1ull));
    }
  }
  if (MINV) {
#pragma unroll
    for (int k = 0; k < kRowFarKeys; ++k) if (k < nk) {
      const int32_t am = q.minv ? q.minv[k] : -1, as = qs.minv ? qs.minv[k] : -1;
      diff |= (uint64_t)(uint32_t)(am ^ v0[k]) & (0ull - (uint64_t)((q.defined >> k) & 1u));
      if (!SAME) diff |= (uint64_t)(uint32_t)(as ^ v1[k]) & (0ull - (uint64_t)((qs.defined >> k) & 1u));
    }
  }
  if (MINV) for (int k = kRowFarKeys; k < nk; ++k) {
    const int32_t am = q.minv ? q.minv[k] : -1, as = qs.minv ? qs.minv[k] : -1;
    diff |= (uint64_t)(uint32_t)(am ^ (bm0 ? bm0[k] : -1)) & (0ull - (uint64_t)((q.defined >> k) & 1u));
    if (!SAME) diff |= (uint64_t)(uint32_t)(as ^ (bm1 ? bm1[k] : -1)) & (0ull - (uint64_t)((qs.defined >> k) & 1u));
  }
  if (q.has_gte | q.has_lte | qs.has_gte | qs.has_lte) {   // bounds: rare, key by key
    const ReqRef b = a.reqs.at(d, rep), bs = a.strict.at(d, rep);
    for (int k = 0; k < nk; ++k) {
      if (bit(q.has_gte, k) && q.gte[k] != b.gte[k]) diff |= 1;
      if (bit(q.has_lte, k) && q.lte[k] != b.lte[k]) diff |= 1;
      if (!SAME && bit(qs.has_gte, k) && qs.gte[k] != bs.gte[k]) diff |= 1;
      if (!SAME && bit(qs.has_lte, k) && qs.lte[k] != bs.lte[k]) diff |= 1;
    }
  }
  if (a.host_ports) diff |= (a.host_ports[(size_t)row * 2] ^ a.host_ports[(size_t)rep * 2]) | (a.host_ports[(size_t)row * 2 + 1] ^ a.host_ports[(size_t)rep * 2 + 1]);
  if (a.vol) diff |= a.vol[row] ^ a.vol[rep];
  if (a.topo_owned) for (int w = 0; w < a.topo_words; ++w) {
    diff |= a.topo_owned[(size_t)row * a.topo_words + w] ^ a.topo_owned[(size_t)rep * a.topo_words + w];
    diff |= a.topo_selected[(size_t)row * a.topo_words + w] ^ a.topo_selected[(size_t)rep * a.topo_words + w];
  }
  return diff;
}
KS_FN void row_hash_body(int row, const RowArgs& a) {
  const ReqRef q = a.reqs.at(a.dict, row), qs = a.strict.at(a.dict, row);
  auto req_at = [&](int r) { return a.requests[(size_t)r * a.n_rows + row]; };
  const uint64_t tol = a.tolerates[row];
  uint32_t slot = 0;
  const uint32_t other = row_table_insert(row, a, row_hash_kept(a, row_hash_value_with(row, a, q, qs, req_at, tol)), &slot);
  a.row_slot[row] = slot;
  if (other == 0xFFFFFFFFu) return;
  const bool differ = row_diff_far(row, a, other, q, qs, req_at, tol) != 0;
  if (differ) *a.collision = 1;
#if !KS_DEVICE
  if (differ == rows_equal(a, row, (int)other)) *a.collision = 2;   // test emulation: the batched comparison against the plain definition, on every comparison of every test
#endif
}
KS_FN void row_class_body(int row, const RowArgs& a) {
  uint32_t slot = a.row_slot[row];
  uint32_t id = a.table_class[slot];
  a.row_class[row] = id;
  if (a.table_rep[slot] == (uint32_t)row) a.class_rep[id] = (uint32_t)row;
}
KS_FN void copy_reqset(const Dict& d, const MutReqTable& dst, uint32_t di, const ReqRef& s, int lane = 0, int nl = 1) {
  uint64_t* m = dst.mask + (size_t)di * d.req_words;
  for (int w = lane; w < d.req_words; w += nl) m[w] = s.mask[w];
  if (lane == 0) { dst.defined[di] = s.defined; dst.complement[di] = s.complement; dst.has_gte[di] = s.has_gte; dst.has_lte[di] = s.has_lte; }
  for (int k = lane; k < d.n_keys; k += nl) {
    dst.gte[(size_t)di * d.n_keys + k] = (s.gte && bit(s.has_gte, k)) ? s.gte[k] : 0;
    dst.lte[(size_t)di * d.n_keys + k] = (s.lte && bit(s.has_lte, k)) ? s.lte[k] : 0;
    dst.minv[(size_t)di * d.n_keys + k] = s.minv ? s.minv[k] : -1;
  }
}
// class tables from the representative row; `lane` of `nl` cooperating lanes (the device runs one wavefront per class, the
// host emulation one call per class)
KS_FN void class_gather_body(int cls, const RowArgs& a, int lane = 0, int nl = 1) {
  int row = (int)a.class_rep[cls];
  for (int r = lane; r < a.n_res; r += nl) {
    int64_t v = a.requests[(size_t)r * a.n_rows + row];
    a.cls_requests[(size_t)cls * a.n_res + r] = v;
    if (v < ((volatile int64_t*)a.min_request)[r]) atomic_min_i64(&a.min_request[r], v);   // a few thousand classes, n_res addresses: read first
  }
  copy_reqset(a.dict, a.cls_reqs, cls, a.reqs.at(a.dict, row), lane, nl);
  copy_reqset(a.dict, a.cls_strict, cls, a.strict.at(a.dict, row), lane, nl);
  if (lane == 0) {
    a.cls_tolerates[cls] = a.tolerates[row];
    if (a.host_ports) { a.cls_host_ports[(size_t)cls * 2] = a.host_ports[(size_t)row * 2]; a.cls_host_ports[(size_t)cls * 2 + 1] = a.host_ports[(size_t)row * 2 + 1]; }
    if (a.vol) a.cls_vol[cls] = a.vol[row];
  }
  // packed records for the pack engine (RecLayout)
  const RecLayout& ly = a.lay;
  ReqRef q = a.reqs.at(a.dict, row);
  uint64_t* hot = a.cls_hot + (size_t)cls * ly.k_hot_words();
  uint64_t* cold = a.cls_cold + (size_t)cls * ly.cold_words();
  for (int w = lane; w < ly.rw; w += nl) hot[ly.k_mask() + w] = q.mask[w];
  for (int r = lane; r < a.n_res; r += nl) hot[ly.k_req() + r] = (uint64_t)a.requests[(size_t)r * a.n_rows + row];
  bool has_minv = false;
  int64_t* cg = (int64_t*)cold; int64_t* cl = cg + ly.nk; int32_t* cv = (int32_t*)(cold + 2 * ly.nk);
  for (int k = 0; k < ly.nk; ++k) {
    int32_t mv = q.minv ? q.minv[k] : -1;
    if (mv >= 0 && bit(q.defined, k)) has_minv = true;
    if (k % nl != lane) continue;
    cg[k] = (q.gte && bit(q.has_gte, k)) ? q.gte[k] : 0;
    cl[k] = (q.lte && bit(q.has_lte, k)) ? q.lte[k] : 0;
    cv[k] = mv;
  }
  if (lane == 0) {
    hot[ly.k_f0()] = (uint64_t)q.defined | ((uint64_t)q.complement << 32);
    hot[ly.k_f1()] = (uint64_t)q.has_gte | ((uint64_t)q.has_lte << 32);
    hot[ly.k_tol()] = a.tolerates[row];
    hot[ly.k_meta()] = has_minv ? 1u : 0u;
  }
  if (a.cls_topo) for (int w = lane; w < a.topo_words; w += nl) {
    a.cls_topo[(size_t)cls * 2 * a.topo_words + w] = a.topo_owned[(size_t)row * a.topo_words + w];
    a.cls_topo[(size_t)cls * 2 * a.topo_words + a.topo_words + w] = a.topo_selected[(size_t)row * a.topo_words + w];
  }
}

// ------------------------------------------------------------------------------------------------ queue order
struct SortKeyArgs {
  int n_pods, n_rows;
  const int64_t* requests;   // [n_res][n_rows]; dim 0 = cpu, dim 1 = memory
  const int64_t* creation;
  const uint64_t *uid_hi, *uid_lo;
  const uint32_t* idx_in;    // current permutation
  uint64_t* key_out;
  int pass;                  // 0 uid_lo, 1 uid_hi, 2 creation, 3 memory (desc), 4 cpu (desc)
};
KS_FN void sort_key_body(int i, const SortKeyArgs& a) {
  uint32_t p = a.idx_in ? a.idx_in[i] : (uint32_t)i;
  uint64_t k;
  switch (a.pass) {
    case 0: k = a.uid_lo[p]; break;
    case 1: k = a.uid_hi[p]; break;
    case 2: k = (uint64_t)a.creation[p] ^ 0x8000000000000000ull; break;
    case 3: k = ~((uint64_t)a.requests[(size_t)1 * a.n_rows + p] ^ 0x8000000000000000ull); break;
    default: k = ~((uint64_t)a.requests[(size_t)0 * a.n_rows + p] ^ 0x8000000000000000ull); break;
  }
  a.key_out[i] = k;
}

// ------------------------------------------------------------------------------------------------ finalize
struct FinalizeArgs {
  Dict dict;
  int n_its, it_words, n_zones, n_cts;
  const uint64_t* it_off_avail;
  const double* it_off_price;
  const uint64_t* c_hot;
  const uint64_t* c_cold;
  RecLayout lay;
  double* cheapest;
  // addDaemonRequests — nodeclaim.go:353-377
  const int* dg_first;
  const int64_t* dg_ov;
  const uint64_t* dg_its;
  uint64_t dg_nonempty;
  const uint64_t* t_its;     // [n_templates][it_words] prefiltered instance types: daemonOverheadGroups are built over them, in their order
  int64_t* daemon_requests;  // [n_claims][n_res]
  // reserved offerings: a claim that holds reservations launches only into them (nodeclaim.go:391-403)
  const uint64_t* c_reserved;    // [n_claims] or nullptr
  const uint32_t* it_resv_first;
  const uint8_t *resv_zone, *resv_id;
  const double* resv_price;
  // Results.TruncateInstanceTypes (scheduler.go:419-437)
  int truncate_n, best_effort;
  ReqTable it_reqs;
  int32_t* sort_idx;         // [n_claims][n_its] the claim's instance types, OrderByPrice order
  double* sort_price;        // [n_claims][n_its]
  uint32_t* ordered_count;   // [n_claims]
  uint8_t* trunc_failed;     // [n_claims]
  // a sweep finalizes the claims of all its probes in one launch: claim c is the record at slot_of[c] of the claim arrays
  // (c_hot, c_cold, c_reserved), its template's prefiltered types are the same for every probe; outputs are indexed by c
  const uint32_t* slot_of;   // null: c
};
// one thread per claim: min over InstanceTypeOptions of the cheapest available offering compatible with the claim's
// requirements (the comparator key of OrderByPrice, types.go:336-355)
KS_FN void finalize_body(int c, const FinalizeArgs& a) {
  const Dict& d = a.dict;
  const RecLayout& ly = a.lay;
  const size_t src = a.slot_of ? (size_t)a.slot_of[c] : (size_t)c;
  const uint64_t* hot = a.c_hot + src * ly.c_hot_words();
  const uint64_t* cold = a.c_cold + src * ly.cold_words();
  ReqRef r;
  r.mask = hot + ly.c_mask();
  r.defined = (uint32_t)hot[ly.c_f0()]; r.complement = (uint32_t)(hot[ly.c_f0()] >> 32);
  r.has_gte = (uint32_t)hot[ly.c_f1()]; r.has_lte = (uint32_t)(hot[ly.c_f1()] >> 32);
  r.gte = (const int64_t*)cold; r.lte = (const int64_t*)(cold + ly.nk); r.minv = nullptr;
  uint32_t zones = 0, cts = 0;
  if (d.key_zone >= 0 && bit(r.defined, d.key_zone)) { for (int z = 0; z < a.n_zones; ++z) if (req_has(d, r, d.key_zone, d.key_word_off[d.key_zone], z)) zones |= 1u << z; }
  else zones = (1u << a.n_zones) - 1;
  if (d.key_ct >= 0 && bit(r.defined, d.key_ct)) { for (int t = 0; t < a.n_cts; ++t) if (req_has(d, r, d.key_ct, d.key_word_off[d.key_ct], t)) cts |= 1u << t; }
  else cts = (1u << a.n_cts) - 1;
  uint64_t cells = 0;
  for (uint32_t zz = zones; zz; zz &= zz - 1) cells |= (uint64_t)cts << (__builtin_ctz(zz) * 4);
  double best = 1.7976931348623157e308;
  const uint64_t* its = hot + ly.c_its();
  for (int w = 0; w < a.it_words; ++w) {
    uint64_t m = its[w];
    while (m) {
      int b = ctz64(m);
      m &= m - 1;
      int it = w * 64 + b;
      uint64_t av = a.it_off_avail[it] & cells;
      while (av) {
        int cell = ctz64(av);
        av &= av - 1;
        double p = a.it_off_price[(size_t)it * 64 + cell];
        if (p < best) best = p;
      }
    }
  }
  if (a.c_reserved && a.c_reserved[src]) {
    const uint64_t held = a.c_reserved[src];
    best = 1.7976931348623157e308;
    for (int w = 0; w < a.it_words; ++w) for (uint64_t m = its[w]; m; m &= m - 1) {
      const int it = w * 64 + ctz64(m);
      for (uint32_t o = a.it_resv_first[it]; o < a.it_resv_first[it + 1]; ++o)
        if (((held >> a.resv_id[o]) & 1) && ((zones >> a.resv_zone[o]) & 1) && a.resv_price[o] < best) best = a.resv_price[o];
    }
  }
  a.cheapest[c] = best;
  if (a.truncate_n > 0) {
    // InstanceTypeOptions.Truncate (types.go:437-449): OrderByPrice — Go's sort.Slice on the cheapest compatible available
    // offering of each type, an unstable sort whose tie order is part of the result — then the first truncate_n types,
    // and minValues must still hold for them (unless the policy is BestEffort).
    int32_t* idx = a.sort_idx + (size_t)c * a.n_its;
    double* pr = a.sort_price + (size_t)c * a.n_its;
    const uint64_t held = (a.c_reserved ? a.c_reserved[src] : 0ull);
    int n = 0;
    for (int w = 0; w < a.it_words; ++w) for (uint64_t m = its[w]; m; m &= m - 1) {
      const int it = w * 64 + ctz64(m);
      double p = 1.7976931348623157e308;
      if (held) {
        for (uint32_t o = a.it_resv_first[it]; o < a.it_resv_first[it + 1]; ++o)
          if (((held >> a.resv_id[o]) & 1) && ((zones >> a.resv_zone[o]) & 1) && a.resv_price[o] < p) p = a.resv_price[o];
      } else {
        for (uint64_t av = a.it_off_avail[it] & cells; av; av &= av - 1) { const double q = a.it_off_price[(size_t)it * 64 + ctz64(av)]; if (q < p) p = q; }
      }
      idx[n] = it; pr[n] = p; n++;
    }
    go_sort_slice(n, [pr](int i, int j) { return pr[i] < pr[j]; },
                  [pr, idx](int i, int j) { const double tp = pr[i]; pr[i] = pr[j]; pr[j] = tp; const int32_t ti = idx[i]; idx[i] = idx[j]; idx[j] = ti; });
    const int keep = n < a.truncate_n ? n : a.truncate_n;
    bool failed = false;
    const uint32_t fl = (uint32_t)(hot[ly.c_meta2()] >> 32);
    if ((fl & 2u) && !a.best_effort) {
      const int32_t* mv = (const int32_t*)(cold + 2 * ly.nk);
      for (int k = 0; k < ly.nk; ++k) {
        if (mv[k] < 0) continue;
        int have = 0;
        if (k == d.key_it) have = keep;
        else for (uint32_t x = d.key_word_off[k]; x < d.key_word_off[k + 1]; ++x) {
          uint64_t u = 0;
          for (int i = 0; i < keep; ++i) u |= a.it_reqs.mask[(size_t)idx[i] * d.req_words + x];
          have += popc64(u);
        }
        if (have < mv[k]) failed = true;
      }
    }
    a.trunc_failed[c] = failed ? 1 : 0;
    a.ordered_count[c] = (uint32_t)(failed ? n : keep);
  }
  // The smallest daemon overhead over the groups that still have an instance type on the claim. Groups are visited in
  // the order NewScheduler built them (first appearance over the template's prefiltered types, scheduler.go:985-1003;
  // the reference's own order is a Go map iteration). MinResources keeps the intersection of keys: a group without any
  // daemon pod empties the running minimum, and an empty minimum is overwritten by the next group (nodeclaim.go:368-372).
  {
    const int nr = ly.nr, iw = a.it_words;
    const int t = (int)((uint32_t)hot[ly.c_meta()] & 31u);
    const int g0 = a.dg_first[t], g1 = a.dg_first[t + 1];
    int64_t cur[kMaxRes];
    bool cur_empty = true;
    uint64_t visited = 0;
    for (int step = g0; step < g1; ++step) {
      int pick = -1, pick_first = 0x7FFFFFFF;
      for (int g = g0; g < g1; ++g) {
        if ((visited >> (g - g0)) & 1) continue;
        int first = 0x7FFFFFFE;
        for (int w = 0; w < iw; ++w) { uint64_t m = a.dg_its[(size_t)g * iw + w] & a.t_its[(size_t)t * iw + w]; if (m) { first = w * 64 + ctz64(m); break; } }
        if (first < pick_first) { pick_first = first; pick = g; }
      }
      if (pick < 0) break;
      visited |= 1ull << (pick - g0);
      bool remaining = false;
      for (int w = 0; w < iw; ++w) if (a.dg_its[(size_t)pick * iw + w] & its[w]) remaining = true;
      if (!remaining) continue;
      const bool g_empty = !((a.dg_nonempty >> pick) & 1);
      if (cur_empty) { for (int r = 0; r < nr; ++r) cur[r] = a.dg_ov[(size_t)pick * nr + r]; cur_empty = g_empty; }
      else { for (int r = 0; r < nr; ++r) { int64_t v = a.dg_ov[(size_t)pick * nr + r]; cur[r] = v < cur[r] ? v : cur[r]; } cur_empty = g_empty; }
    }
    for (int r = 0; r < nr; ++r) a.daemon_requests[(size_t)c * nr + r] = cur_empty ? 0 : cur[r];
  }
}


// ------------------------------------------------------------------------------------------------ resident-cluster sweeps
// ksolve_node_dead0: every pod class against every PRISTINE existing node of a resident cluster, once per base handle, for
// all the probes of every sweep (disruption/helpers.go:53-155 runs Solve() once per candidate set against the same cluster).
// dead0[k][w] bit b = node 64w+b fails ExistingNode.CanAdd for class k before volume alternatives / topology
// (existingnode.go:81-106; nodecheck.h). One wavefront per 64 consecutive nodes, the classes one after the other: the node
// side is read coalesced (SoA tables, lane = node), the class record is the same address for every lane.
struct NodeDeadArgs {
  Dict dict;
  RecLayout lay;
  int n_nodes, node_words, n_classes, hp_on;
  const uint64_t* cls_hot;      // [n_classes][k_hot_words]
  const uint64_t* cls_cold;     // [n_classes][cold_words]
  const uint64_t* cls_hp;       // [n_classes][2] or null
  const uint64_t* node_taints;  // [n_nodes]
  NodeTabs pristine;
  uint64_t* dead0;              // [n_classes][node_words]
};
// One wavefront per (64 nodes, kDead0Classes classes): every lane reads its node's scalars ONCE into registers (taints, host ports,
// remaining resources, defined / complement / bound flags) and tests the chunk's classes against them; only the requirement-mask
// words of the keys a class and the node both define are read per class (64 lanes x 8 B: one coalesced 512-byte access from L1).
// Round 5's form — one wavefront per 64 nodes looping over ALL classes, its node tables re-read per class because the store of
// the result word may alias them — kept 1,563 wavefronts on 1,024 SIMDs busy for 3.4 ms at 100k nodes x 1,182 classes: 1.5
// wavefronts per SIMD, each a chain of dependent loads. The class chunks make it ~58,000 wavefronts.
constexpr int kDead0Classes = 32;
template <class W>
KS_DEV void node_dead0_body(int block, int chunk, const NodeDeadArgs& a) {
  const int base = block * 64;
  const int cnt = a.n_nodes - base < 64 ? a.n_nodes - base : 64;
  const RecLayout ly = a.lay;
  LaneVar<uint64_t> tv, hpv;
  LaneVar<int64_t> r0, r1, r2, r3, r4, r5, r6, r7;
  LaneVar<uint32_t> dv, cv, gv, lv;
  W::each([&](int l) {
    const size_t i = (size_t)(base + (l < cnt ? l : cnt - 1));
    const NodePre n = node_preload(ly, a.node_taints[i], a.pristine, i);
    tv.at(l) = n.taints; hpv.at(l) = n.hp;
    r0.at(l) = n.rem[0]; r1.at(l) = n.rem[1]; r2.at(l) = n.rem[2]; r3.at(l) = n.rem[3]; r4.at(l) = n.rem[4]; r5.at(l) = n.rem[5]; r6.at(l) = n.rem[6]; r7.at(l) = n.rem[7];
    dv.at(l) = n.ndef; cv.at(l) = n.ncomp; gv.at(l) = n.nhg; lv.at(l) = n.nhl;
  });
  const int k0 = chunk * kDead0Classes, k1 = k0 + kDead0Classes < a.n_classes ? k0 + kDead0Classes : a.n_classes;
  for (int k = k0; k < k1; ++k) {
    const uint64_t* cls = a.cls_hot + (size_t)k * ly.k_hot_words();
    const NodeClassCtx cx = node_class_ctx(a.dict, ly, cls, a.cls_cold + (size_t)k * ly.cold_words(), (a.hp_on && a.cls_hp) ? a.cls_hp[(size_t)k * 2 + 1] : 0ull);
    const uint64_t ok = W::ballot([&](int l) {
      NodePre n;
      n.taints = tv.at(l); n.hp = hpv.at(l);
      n.rem[0] = r0.at(l); n.rem[1] = r1.at(l); n.rem[2] = r2.at(l); n.rem[3] = r3.at(l); n.rem[4] = r4.at(l); n.rem[5] = r5.at(l); n.rem[6] = r6.at(l); n.rem[7] = r7.at(l);
      n.ndef = dv.at(l); n.ncomp = cv.at(l); n.nhg = gv.at(l); n.nhl = lv.at(l);
      return l < cnt && node_static_ok_pre(a.dict, ly, cx, n, a.pristine, (size_t)(base + (l < cnt ? l : cnt - 1)));
    });
    W::store(&a.dead0[(size_t)k * a.node_words + block], (uint64_t)~ok);
  }
}

// the claims a sweep's probes created, gathered into compact arrays for the download (one thread per claim)
struct ClaimGatherArgs {
  RecLayout lay;
  const uint32_t* slot_of;     // [n] slot in the sweep's claim arrays
  const uint64_t *c_hot, *c_cold, *c_reserved;
  uint64_t *out_hot, *out_cold, *out_reserved;
};
KS_FN void claim_gather_body(int i, const ClaimGatherArgs& a) {
  const size_t src = a.slot_of[i];
  const int hw = a.lay.c_hot_words(), cw = a.lay.cold_words();
  for (int w = 0; w < hw; ++w) a.out_hot[(size_t)i * hw + w] = a.c_hot[src * hw + w];
  for (int w = 0; w < cw; ++w) a.out_cold[(size_t)i * cw + w] = a.c_cold[src * cw + w];
  a.out_reserved[i] = a.c_reserved[src];
}

// A sweep's workspace records, built ON THE DEVICE (round 6). A probe's Workspace is ~600 bytes of pointers into the sweep's arena; built
// on the host and uploaded they were 6 MB per 10,000 probes — most of the call's upload phase on a host link that moves 3.5 GB/s.
// The host now sends one 48-byte descriptor per probe (where its block of the arena starts, its pods / removed nodes / claim slots)
// and this function — the SAME code the host runs to measure a probe's block (arena = null) — lays the record out on the device,
// one thread per probe (ksolve_sweep_items). Every region is rounded up to 64 bytes as the arena's take() rounds it.
struct SweepItemDesc { uint64_t off, hl_off; uint32_t pod_b, m, node_b, n_removed, cb, mc, pv_entries, pad; };
static_assert(sizeof(SweepItemDesc) == 48, "SweepItemDesc layout");
struct SweepItemArgs {
  char* arena;                    // null: measure only (the pointers written are meaningless, the sizes returned are exact)
  const SweepItemDesc* desc;      // [n]
  Workspace* items;               // [n] out
  volatile int* const* cancel_each;   // [n] or null: the flag probe i polls where it has one of its own (probe handles in a batch)
  // regions every probe has a share of
  uint32_t *d_sorted, *d_removed, *d_slot, *d_last, *d_queue, *d_okey, *d_oord, *d_opos;
  int64_t* d_limits; int32_t* d_assign; uint8_t *d_err, *d_diag;
  uint64_t *d_hot, *d_cold, *d_resv, *d_chp;
  int *d_nclaims, *d_status; Counters* d_ctr;
  // the base handle's pristine node tables, its cancel flag and options
  const uint64_t* n_mask0; const uint32_t *n_defined0, *n_complement0; const int64_t* n_remaining0;
  volatile int* cancel; long long max_steps; int min_values_best_effort, order_cap;
  // sizes
  uint32_t ne, nr, T, nc, nk, it_words, req_words, hot_words, cold_words;
  uint32_t bounds, has_topology, pv_on, hp_on;
  uint32_t G, dom_words, hg, n_alias, n_key_slots;
};
// fills *W for probe p; returns the bytes of the probe's own block (from desc.off) and, through hl_bytes, of its hostname-threshold bitmaps
KS_FN size_t sweep_item_fill(Workspace* W, const SweepItemArgs& A, const SweepItemDesc& d, uint32_t p, size_t* hl_bytes) {
  size_t off = d.off;
  char* const arena = A.arena;
  auto take = [&](size_t bytes) -> void* { void* q = arena ? (void*)(arena + off) : nullptr; off += (bytes + 63) & ~(size_t)63; return q; };
  const uint32_t m = d.m, mc = d.mc, cw = (mc + 63) / 64, nr = A.nr, T = A.T;
  uint32_t oc = 64;
  { const uint32_t a = m > 1 ? m : 1, b = A.ne > 1 ? A.ne : 1; while (oc < 2 * (a < b ? a : b)) oc <<= 1; }
  Workspace w = Workspace{};
  w.c_headroom = (int64_t*)take(((size_t)mc * nr + 64) * 8);
  w.t_its = (uint64_t*)take((size_t)T * A.it_words * 8);
  w.t_remaining = (int64_t*)take((size_t)T * (nr + 1) * 8);
  w.dead = (uint64_t*)take((size_t)A.nc * cw * 8);
  if (A.ne) {
    w.ov_key = (uint32_t*)take((size_t)oc * 4); w.pr_revived = (uint32_t*)take((size_t)oc * 4);
    w.n_mask = (uint64_t*)take((size_t)A.req_words * oc * 8);
    w.n_defined = (uint32_t*)take((size_t)oc * 4); w.n_complement = (uint32_t*)take((size_t)oc * 4);
    if (A.bounds) { w.n_hg = (uint32_t*)take((size_t)oc * 4); w.n_hl = (uint32_t*)take((size_t)oc * 4); w.n_gte = (int64_t*)take((size_t)A.nk * oc * 8); w.n_lte = (int64_t*)take((size_t)A.nk * oc * 8); }
    w.n_remaining = (int64_t*)take((size_t)nr * oc * 8);
    w.n_npods = (uint32_t*)take((size_t)oc * 4);
    w.n_hp = A.hp_on ? (uint64_t*)take((size_t)oc * 8) : nullptr;
    if (A.has_topology) {
      const size_t G = A.G, dv = (size_t)A.dom_words * 64, hg = A.hg, ks_ = A.n_key_slots;
      w.tg_domains = (uint64_t*)take(G * A.dom_words * 8); w.tg_counts = (int32_t*)take(G * dv * 4); w.tg_regs = (int32_t*)take(G * dv * 4);
      w.tg_node_counts = (int32_t*)take(hg * oc * 4); w.tg_claim_counts = (int32_t*)take(hg * mc * 4);   // per overlay slot: what this probe's commits add to the cluster's shared per-node counts
      w.tg_nonzero = (int32_t*)take(G * 4); w.tg_alias_active = A.n_alias ? (int32_t*)take((size_t)A.n_alias * 4) : nullptr;
      w.c_keymask = (uint64_t*)take(ks_ * mc * 8);
      w.kv_claims = (uint64_t*)take(ks_ * 64 * cw * 8);
    }
    if (A.pv_on) w.pv_log = (uint64_t*)take((size_t)(d.pv_entries > 1 ? d.pv_entries : 1) * 8);
  }
  w.ov_cap = (int)oc;
  const size_t own = off - d.off;
  size_t hl = 0;
  if (A.has_topology) {   // starts as ones (all claims below every threshold): it lives behind the zero-filled regions
    hl = (((size_t)A.hg * 2 * cw * 8) + 63) & ~(size_t)63;
    w.host_le = arena ? (uint64_t*)(arena + d.hl_off) : nullptr;
  }
  if (hl_bytes) *hl_bytes = hl;
  if (!W) return own;
  const uint32_t b = d.pod_b, cb = d.cb;
  w.max_claims = (int)mc; w.claim_words = (int)cw;
  w.c_hot = A.d_hot + (size_t)cb * A.hot_words; w.c_cold = A.d_cold + (size_t)cb * A.cold_words;
  w.c_reserved = A.d_resv + cb; w.c_hp = A.d_chp ? A.d_chp + cb : nullptr;
  w.o_key = A.d_okey + cb; w.o_ord = A.d_oord + cb; w.o_pos = A.d_opos + cb;
  w.queue = A.d_queue + b + p; w.last_len = A.d_last + b;
  w.assign = A.d_assign + b; w.slot = A.d_slot + b; w.err = A.d_err + b; w.diag = A.d_diag + b;
  w.n_claims_out = A.d_nclaims + p; w.status_out = A.d_status + p; w.counters = A.d_ctr + p;
  w.cancel_flag = (A.cancel_each && A.cancel_each[p]) ? A.cancel_each[p] : A.cancel;   // ksolve_cancel(base) stops every probe of a ksolve_sweep
  w.max_steps = A.max_steps;
  w.min_values_best_effort = A.min_values_best_effort;
  w.n_mask0 = A.n_mask0; w.n_defined0 = A.n_defined0; w.n_complement0 = A.n_complement0; w.n_remaining0 = A.n_remaining0;
  w.probe = 1; w.pr_n_pods = (int)m; w.pr_sorted = A.d_sorted + b;
  w.pr_removed = A.d_removed + d.node_b; w.pr_n_removed = (int)d.n_removed;
  w.pr_limits = A.d_limits ? A.d_limits + (size_t)p * T * (nr + 1) : nullptr;
  w.pr_order_cap = A.order_cap;
  *W = w;
  return own;
}
KS_FN void sweep_items_body(int i, const SweepItemArgs& a) { sweep_item_fill(a.items + i, a, a.desc[i], (uint32_t)i, nullptr); }

}  // namespace ks

