// run_order.h — the claim order of the BIG engine (more in-flight claims than a CU's LDS holds): the same permutation as
// pdq_emul.h's ClaimOrder — Go's sort.Slice(newNodeClaims, by pod count) before every in-flight scan (scheduler.go:598) —
// kept as ONE RING PER POD COUNT instead of one array.
//
// Why: with the order in an array, a claim that gains a pod moves to the end of its run of equally full claims, and
// everything behind it in that run steps one position to the left. With tens of thousands of equally full claims (every
// anti-affinity pod of BASELINE configs[2] is a claim of its own) that is tens of thousands of elements per pod, each with
// a scattered write of its inverse entry: O(claims) per pod, 65% of the solve at 200k claims.
//
// The array is always "sorted by count except for the claim the last step touched", and pdqsort repairs that with one
// stable move in all but a few cases (see pdq_emul.h): the claim leaves its run — anywhere — and becomes the FIRST element
// of the next run; a new claim becomes the LAST element of run 1. With one ring per count
//     position(claim) = prefix[count] + ((slot[claim] - head[count]) mod capacity[count])
// both are O(1): pop (the shorter side of the ring steps over the hole), push-front / push-back, and prefix[count + 1]
// loses one (the only prefix that changes). Every other path of pdqsort (a pivot sample that sees the defect, 12 < n < 50)
// materialises the array, runs the literal emulation on it (ClaimOrder::sort) and rebuilds the rings: O(n), a few hundred
// times per million pods.
#pragma once
#include "pdq_emul.h"

namespace ks {

constexpr int kRunMaxCount = 1024;   // counts whose table entries are mirrored in LDS; fuller claims use the HBM tables alone

// LDS mirror of the tables' first kRunMaxCount entries, one 16-byte record per count (the spread engine reads the records of the
// sixteen lowest runs with ONE ds_read_b128, one lane each: topo_engine.h window())
struct RunEnt {
  uint32_t head;     // ring index of the run's first claim
  uint32_t size;     // claims in the run
  uint32_t prefix;   // claims with a smaller count = position of the run's first claim
  uint32_t off;      // the ring's first word in `ring`
};
struct RunTables {
  alignas(16) RunEnt e[kRunMaxCount];
  uint8_t log2cap[kRunMaxCount];   // ring capacity = 1 << log2cap
};

template <class W>
struct RunOrder {
  KS_LDS RunTables* T = nullptr;
  // HBM tables, [kmax] each (kmax = pods of the problem + 2: a claim cannot hold more); entries below kRunMaxCount are
  // read from the LDS mirror
  uint32_t* ghead = nullptr; uint32_t* gsize = nullptr; uint32_t* gprefix = nullptr;
  const uint32_t* goff = nullptr; const uint8_t* glog = nullptr;
  int kmax = 0;
  uint32_t* ring = nullptr;   // HBM: all rings
  uint32_t* cnt = nullptr;    // HBM [max_claims]: pod count of the claim (the run it is in)
  uint32_t* slot = nullptr;   // HBM [max_claims]: ring index of the claim inside its run
  uint32_t* key = nullptr;    // HBM [max_claims]: array form, materialised on demand
  uint32_t* ord = nullptr;
  int n = 0;                  // claims, including a pending new one
  int max_cnt = 1;            // largest count a claim has
  // the claim the last step touched; its move happens in sort()
  int defect = -1;            // its position, -1: none
  int defect_claim = -1;
  bool defect_append = false;
  bool overflow = false;
  uint64_t slow_sorts = 0;

  KS_DEV void init(KS_LDS RunTables* t, uint32_t* ring_, uint32_t* cnt_, uint32_t* slot_, uint32_t* key_, uint32_t* ord_, uint32_t* tabs, const uint32_t* off_tab, const uint8_t* log_tab, int kmax_) {
    T = t; ring = ring_; cnt = cnt_; slot = slot_; key = key_; ord = ord_; kmax = kmax_;
    ghead = tabs; gsize = tabs + kmax_; gprefix = tabs + 2 * (size_t)kmax_; goff = off_tab; glog = log_tab;
    KS_LDS RunTables* tt = t;
    const int km = kmax_;
    W::for_n(kRunMaxCount, [&](int k) { tt->e[k].head = 0; tt->e[k].size = 0; tt->e[k].prefix = 0; tt->e[k].off = k < km ? off_tab[k] : 0; tt->log2cap[k] = k < km ? log_tab[k] : 0; });
    // the HBM tables above the mirror are cleared as the largest count grows (grow_to)
    n = 0; max_cnt = 1; defect = -1; defect_claim = -1; defect_append = false; overflow = false; slow_sorts = 0;
  }
  // table entries: LDS below kRunMaxCount, HBM above
  KS_FN uint32_t head_(int k) const { return k < kRunMaxCount ? T->e[k].head : ghead[k]; }
  KS_FN uint32_t size_(int k) const { return k < kRunMaxCount ? T->e[k].size : gsize[k]; }
  KS_FN uint32_t prefix_(int k) const { return k < kRunMaxCount ? T->e[k].prefix : gprefix[k]; }
  KS_FN uint32_t off_(int k) const { return k < kRunMaxCount ? T->e[k].off : goff[k]; }
  KS_FN uint32_t mask_of(int k) const { return (1u << (k < kRunMaxCount ? T->log2cap[k] : glog[k])) - 1u; }
  KS_DEV void set_head(int k, uint32_t v) { if (k < kRunMaxCount) W::store(&T->e[k].head, v); else W::store(&ghead[k], v); }
  KS_DEV void set_size(int k, uint32_t v) { if (k < kRunMaxCount) W::store(&T->e[k].size, v); else W::store(&gsize[k], v); }
  KS_DEV void set_prefix(int k, uint32_t v) { if (k < kRunMaxCount) W::store(&T->e[k].prefix, v); else W::store(&gprefix[k], v); }
  // a claim reaches count k for the first time: run k and the prefix behind it start out empty / "every claim"
  KS_DEV void grow_to(int k, uint32_t claims_before_next) {
    if (k <= max_cnt) return;
    set_head(k, 0); set_size(k, 0);
    if (k + 1 < kmax) set_prefix(k + 1, claims_before_next);
    max_cnt = k;
    W::sync();
  }

  // ---- lookups (any lane, any argument) ----
  KS_FN uint32_t position(int c) const {
    if (defect_append && c == defect_claim) return (uint32_t)(n - 1);   // a new claim sits at the end until the next sort
    const int k = (int)cnt[c];
    return prefix_(k) + ((slot[c] - head_(k)) & mask_of(k));
  }
  KS_FN uint32_t claim_at(int p) const {
    if (defect_append && p == n - 1) return (uint32_t)defect_claim;
    // the run that holds position p: the last count whose prefix is <= p (empty runs share their successor's prefix)
    int lo = 1, hi = max_cnt;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (prefix_(mid) <= (uint32_t)p) lo = mid; else hi = mid - 1; }
    return ring[off_(lo) + ((head_(lo) + ((uint32_t)p - prefix_(lo))) & mask_of(lo))];
  }

  // ---- mutation by the scheduler ----
  KS_DEV void increment(int claim) {   // a pod was added to an in-flight claim (nodeclaim.go:249): it moves at the next sort
    defect = (int)W::uniform((uint64_t)position(claim)); defect_claim = claim; defect_append = false;
  }
  KS_DEV void append(int claim) {      // a new claim with its first pod (scheduler.go:785): at the end until the next sort
    W::store(&cnt[claim], 1u);
    W::sync();
    defect = n; defect_claim = claim; defect_append = true;
    n++;
  }

  // the run's elements [from, to) step by `delta` ring slots (+1 / -1); at most the shorter side of the run
  KS_DEV void shift_ring(int k, uint32_t from, uint32_t to, int delta) {
    const uint32_t m = mask_of(k), h = head_(k);
    uint32_t* r = ring + off_(k);
    uint32_t* sl = slot;
    const int len = (int)(to - from);
    // rounds of 64 x 16 elements, read before written; going away from the hole so that no round reads what an earlier wrote
    constexpr int kPer = 16, kRound = 64 * kPer;
    for (int done = 0; done < len; done += kRound) {
      const int cntr = len - done < kRound ? len - done : kRound;
      // delta = +1: the elements before the hole, last first; delta = -1: the elements behind the hole, first first
      const uint32_t base = delta > 0 ? to - (uint32_t)done - (uint32_t)cntr : from + (uint32_t)done;
#if KS_DEVICE
      uint32_t v[kPer];
#pragma unroll
      for (int j = 0; j < kPer; ++j) { const int i = j * 64 + W::lane(); if (i < cntr) v[j] = r[(h + base + (uint32_t)i) & m]; }
      W::sync();
#pragma unroll
      for (int j = 0; j < kPer; ++j) { const int i = j * 64 + W::lane(); if (i < cntr) { const uint32_t s = (h + base + (uint32_t)i + (uint32_t)delta) & m; r[s] = v[j]; sl[v[j]] = s; } }
      W::sync();
#else
      if (delta > 0) for (int i = cntr - 1; i >= 0; --i) { const uint32_t x = r[(h + base + (uint32_t)i) & m]; const uint32_t s = (h + base + (uint32_t)i + 1u) & m; r[s] = x; sl[x] = s; }
      else for (int i = 0; i < cntr; ++i) { const uint32_t x = r[(h + base + (uint32_t)i) & m]; const uint32_t s = (h + base + (uint32_t)i - 1u) & m; r[s] = x; sl[x] = s; }
#endif
    }
  }
  // the single stable move of pdqsort's partialInsertionSort / insertionSort on "sorted except one element"
  KS_DEV void apply_move() {
    const int x = defect_claim;
    if (defect_append) { move_appended(x); return; }
    const int k = (int)W::uniform((uint64_t)cnt[x]);
    if (k + 2 >= kmax) { overflow = true; return; }
    const uint32_t i = (uint32_t)W::uniform((uint64_t)((slot[x] - head_(k)) & mask_of(k)));   // index inside the run
    move_known(x, k, i);
  }
  // a new claim (count 1, at the end of the array): behind the last claim with at most one pod — the end of run 1; every later run
  // starts one position further right
  KS_DEV void move_appended(int x) {
    const uint32_t m = mask_of(1), s = (head_(1) + size_(1)) & m;
    W::store(&ring[off_(1) + s], (uint32_t)x);
    W::store(&slot[x], s);
    set_size(1, size_(1) + 1);
    KS_LDS RunTables* tt = T;
    uint32_t* gp = gprefix;
    const int top = max_cnt + 1;   // prefix is kept for counts 1 .. max_cnt + 1
    W::for_n(top - 1, [&](int i) { const int k = i + 2; if (k < kRunMaxCount) tt->e[k].prefix += 1; else gp[k] += 1; });
  }
  // claim x, the i-th of run k, has gained a pod: it leaves its run and becomes the first claim of run k + 1 (callers that know k and
  // i — the spread engine has both from its scan — skip the two dependent loads of apply_move)
  KS_DEV void move_known(int x, int k, uint32_t i) {
    if (k + 2 >= kmax) { overflow = true; return; }
    grow_to(k + 1, (uint32_t)n);
    const uint32_t m = mask_of(k), h = head_(k), sz = size_(k);
    uint32_t new_head = h;
    if (i == 0) new_head = (h + 1) & m;
    else if (i + 1 < sz) {
      if (i <= sz - 1 - i) { shift_ring(k, 0, i, +1); new_head = (h + 1) & m; }   // the claims before it step towards the hole
      else shift_ring(k, i + 1, sz, -1);                                          // the claims behind it do
    }
    // first element of the next run
    const uint32_t m1 = mask_of(k + 1), h1 = (head_(k + 1) - 1u) & m1;
    W::store(&ring[off_(k + 1) + h1], (uint32_t)x);
    W::store(&slot[x], h1);
    W::store(&cnt[x], (uint32_t)(k + 1));
    const uint32_t s1 = size_(k + 1), p1 = prefix_(k + 1);
    set_head(k, new_head); set_size(k, sz - 1); set_head(k + 1, h1); set_size(k + 1, s1 + 1); set_prefix(k + 1, p1 - 1);
    W::sync();
  }
  // would sort() repair a defect at position p with ONE stable move (pdqsort's insertion sort / partialInsertionSort path)?
  KS_FN bool single_move(int p) const {
    if (n <= 12) return true;
    if (n < 50) return false;
    const int q = n / 4;
    return !((p >= q - 1 && p <= q + 1) || (p >= 2 * q - 1 && p <= 2 * q + 1) || (p >= 3 * q - 1 && p <= 3 * q + 1));
  }
  // array form: key[p] / ord[p] for every position, the touched claim still where it was (with its new count)
  KS_DEV void materialize() {
    uint32_t* kk = key; uint32_t* oo = ord; const uint32_t* rr = ring;
    for (int k = 1; k <= max_cnt; ++k) {
      const uint32_t sz = size_(k);
      if (!sz) continue;
      const uint32_t p0 = prefix_(k), h = head_(k), m = mask_of(k), o = off_(k);
      W::for_n((int)sz, [&](int i) { kk[p0 + (uint32_t)i] = (uint32_t)k; oo[p0 + (uint32_t)i] = rr[o + ((h + (uint32_t)i) & m)]; });
    }
    if (defect_claim >= 0) {
      if (defect_append) { W::store(&key[n - 1], 1u); W::store(&ord[n - 1], (uint32_t)defect_claim); }
      else W::store(&key[defect], (uint32_t)(cnt[defect_claim] + 1u));
    }
    W::sync();
  }
  // rings from the (sorted) array form
  KS_DEV void rebuild() {
    const uint32_t* kk = key; const uint32_t* oo = ord;
    const int nn = n;
    const int mx = (int)W::uniform((uint64_t)kk[nn - 1]);   // the array is sorted: its last count is the largest
    if (mx + 2 >= kmax) { overflow = true; return; }
    KS_LDS RunTables* tt = T;
    uint32_t* gh = ghead; uint32_t* gs = gsize; uint32_t* gp = gprefix;
    // prefix[k] = first position whose count is >= k: one binary search per count, counts 1 .. mx + 1
    W::for_n(mx + 1, [&](int k1) {
      const int k = k1 + 1;
      int lo = 0, hi = nn;
      while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (kk[mid] < (uint32_t)k) lo = mid + 1; else hi = mid; }
      if (k < kRunMaxCount) { tt->e[k].prefix = (uint32_t)lo; tt->e[k].head = 0; } else { gp[k] = (uint32_t)lo; gh[k] = 0; }
    });
    W::for_n(mx, [&](int k1) {
      const int k = k1 + 1;
      const uint32_t a = k < kRunMaxCount ? tt->e[k].prefix : gp[k], b = k + 1 < kRunMaxCount ? tt->e[k + 1].prefix : gp[k + 1];
      if (k < kRunMaxCount) tt->e[k].size = b - a; else gs[k] = b - a;
    });
    max_cnt = mx;
    uint32_t* rr = ring; uint32_t* cc = cnt; uint32_t* ss = slot;
    const uint32_t* go = goff;
    W::for_n(nn, [&](int p) {
      const uint32_t k = kk[p], x = oo[p];
      const uint32_t p0 = k < (uint32_t)kRunMaxCount ? tt->e[k].prefix : gp[k], o = k < (uint32_t)kRunMaxCount ? tt->e[k].off : go[k];
      const uint32_t s = (uint32_t)p - p0;
      rr[o + s] = x; cc[x] = k; ss[x] = s;
    });
  }
  // sort.Slice on the current order
  KS_DEV void sort() {
    if (defect_claim < 0) return;
    const int p = defect;
    bool fast = n <= 12;
    if (n >= 50) {
      const int q = n / 4;
      const bool sampled = (p >= q - 1 && p <= q + 1) || (p >= 2 * q - 1 && p <= 2 * q + 1) || (p >= 3 * q - 1 && p <= 3 * q + 1);
      fast = !sampled;
    }
    if (fast) { apply_move(); defect = -1; defect_claim = -1; defect_append = false; return; }
    materialize();
    ClaimOrder<W, uint32_t*, false> t;
    t.key = key; t.ord = ord; t.pos = nullptr; t.n = n; t.defect = defect; t.defect_append = defect_append;
    t.sort();
    slow_sorts += t.slow_sorts;
    defect = -1; defect_claim = -1; defect_append = false;
    rebuild();
  }
  // the final order for the results (the last step's move stays undone, as in the reference)
  KS_DEV void write_final() { if (n > 0) materialize(); }
};

}  // namespace ks
