// pdq_emul.h — bit-exact emulation of the claim ordering the reference maintains with
//     sort.Slice(s.newNodeClaims, func(a, b int) bool { return len(a.Pods) < len(b.Pods) })     scheduler.go:598
// which runs before every in-flight scan and is Go's UNSTABLE pattern-defeating quicksort (go1.26 sort/zsortfunc.go:
// insertion sort <= 12, ninther pivot, partialInsertionSort, partitionEqual, breakPatterns xorshift, heapsort
// fallback). Which of two equally-full claims a pod lands in depends on the permutation that algorithm leaves, so
// the device keeps the claims in exactly that permutation.
//
// The array is always "sorted except for the one claim the previous step touched" (its count went up by one, or it
// was appended with count 1). For that input pdqsort almost always takes the partialInsertionSort path, whose effect
// is a stable move of one element; that case is handled with a vector search + rotate (O(distance/64) wave steps).
// Every other path (pivot samples that see the defect, 12 < n < 50, ...) runs the full algorithm below with its
// sequential scans replaced by ballot searches. Keys are the per-position pod counts; `ord` holds the claim id at
// each position and `pos` its inverse.
#pragma once
#include "wave.h"

namespace ks {

// P32 = pointer type of the three arrays: LDS for problems whose claims fit the CU's LDS, HBM for larger ones.
// POS = false drops the inverse array (the cursor engine always knows the position of the claim it touches).
template <class W, class P32 = KS_LDS uint32_t*, bool POS = true>
struct ClaimOrder {
  P32 key;   // [cap] pod count of the claim at position i
  P32 ord;   // [cap] claim id at position i
  P32 pos;   // [cap] position of claim id
  int n = 0;
  int defect = -1;        // position whose key changed since the array was last sorted, -1 = sorted
  bool defect_append = false;
  uint64_t slow_sorts = 0;

  // ---- lookups (the BIG engine's RunOrder answers the same two questions from its rings) ----
  KS_FN uint32_t position(int c) const { return pos[c]; }
  KS_FN uint32_t claim_at(int i) const { return ord[i]; }
  // ---- element access (uniform) ----
  KS_FN bool less(int i, int j) const { return key[i] < key[j]; }
  KS_FN void swap(int i, int j) {
    if (i == j) return;
    uint32_t ki = key[i], kj = key[j], oi = ord[i], oj = ord[j];
    W::store(&key[i], kj); W::store(&key[j], ki);
    W::store(&ord[i], oj); W::store(&ord[j], oi);
    if constexpr (POS) { W::store(&pos[oj], (uint32_t)i); W::store(&pos[oi], (uint32_t)j); }
    W::sync();
  }
  // move element at `from` to `to` (to < from), shifting [to, from) right by one
  KS_FN void rotate_right(int to, int from) {
    if (to >= from) return;
    uint32_t mk = key[from], mo = ord[from];
    for (int top = from; top > to; top -= kRound) {  // high to low so a round never reads what an earlier round wrote
      int lo = top - kRound > to ? top - kRound : to;
      shift_round(lo, top, +1, false);
    }
    W::store(&key[to], mk); W::store(&ord[to], mo); if constexpr (POS) W::store(&pos[mo], (uint32_t)to);
    W::sync();
  }
  // move element at `from` to `to` (to > from), shifting (from, to] left by one. same_keys: the elements that shift all carry
  // ONE count (the stable move of a claim that gained a pod passes claims of exactly its old count): their keys need not
  // move at all — one key written at each end instead of one per element.
  KS_FN void rotate_left(int from, int to, bool same_keys = false) {
    if (to <= from) return;
    uint32_t mk = key[from], mo = ord[from];
    const uint32_t passed = key[from + 1];
    for (int lo = from + 1; lo <= to; lo += kRound) {
      int hi = lo + kRound <= to + 1 ? lo + kRound : to + 1;
      shift_round(lo, hi, -1, same_keys);
    }
    if (same_keys) W::store(&key[from], passed);
    W::store(&key[to], mk); W::store(&ord[to], mo); if constexpr (POS) W::store(&pos[mo], (uint32_t)to);
    W::sync();
  }
  // elements [lo,hi) (at most kRound) move by delta (+1 / -1): all reads of the round precede its writes, and
  // all of them are in flight together (a long move — tens of thousands of equally full claims — is bound by round trips)
  static constexpr int kPerLane = sizeof(P32) == 8 ? 16 : 1;   // HBM-resident order (64-bit pointers): sixteen elements per lane per round
  static constexpr int kRound = 64 * kPerLane;
  // searches over the order: one round trip per 64 elements in LDS; eight rounds in flight over an HBM-resident order
  template <class F> KS_FN int ff(int lo, int hi, F pred) const { if constexpr (kPerLane > 1) return W::find_first8(lo, hi, pred); else return W::find_first(lo, hi, pred); }
  template <class F> KS_FN int fl(int lo, int hi, F pred) const { if constexpr (kPerLane > 1) return W::find_last8(lo, hi, pred); else return W::find_last(lo, hi, pred); }
  KS_FN void shift_round(int lo, int hi, int delta, bool same_keys) {
#if KS_DEVICE
    uint32_t k[kPerLane], o[kPerLane];
    if (same_keys) {
#pragma unroll
      for (int j = 0; j < kPerLane; ++j) { const int i = lo + j * 64 + W::lane(); if (i < hi) o[j] = ord[i]; }
      W::sync();
#pragma unroll
      for (int j = 0; j < kPerLane; ++j) { const int i = lo + j * 64 + W::lane(); if (i < hi) { ord[i + delta] = o[j]; if constexpr (POS) pos[o[j]] = (uint32_t)(i + delta); } }
      W::sync();
      return;
    }
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) { const int i = lo + j * 64 + W::lane(); if (i < hi) { k[j] = key[i]; o[j] = ord[i]; } }
    W::sync();
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) { const int i = lo + j * 64 + W::lane(); if (i < hi) { key[i + delta] = k[j]; ord[i + delta] = o[j]; if constexpr (POS) pos[o[j]] = (uint32_t)(i + delta); } }
    W::sync();
#else
    if (delta > 0) for (int i = hi - 1; i >= lo; --i) { if (!same_keys) key[i + 1] = key[i]; ord[i + 1] = ord[i]; if constexpr (POS) pos[ord[i + 1]] = i + 1; }
    else for (int i = lo; i < hi; ++i) { if (!same_keys) key[i - 1] = key[i]; ord[i - 1] = ord[i]; if constexpr (POS) pos[ord[i - 1]] = i - 1; }
#endif
  }
  // first x in [lo,hi) with key[x] >= v, hi if none — [lo,hi) is sorted ascending (binary search: a long run of equal keys
  // costs log2 n round trips instead of n/64)
  KS_FN int lower_bound_sorted(int lo, int hi, uint32_t v) const {
    while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (key[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
  }
  // last x in [lo,hi) with key[x] <= v, lo-1 if none — [lo,hi) sorted ascending
  KS_FN int upper_last_sorted(int lo, int hi, uint32_t v) const {
    int a = lo, b = hi;
    while (a < b) { const int mid = a + (b - a) / 2; if (!(v < key[mid])) a = mid + 1; else b = mid; }
    return a - 1;
  }

  // ---- mutation by the scheduler ----
  KS_FN void increment(int claim) {  // a pod was added to an in-flight claim (nodeclaim.go:249)
    int p = (int)pos[claim];
    W::store(&key[p], key[p] + 1);
    W::sync();
    defect = p; defect_append = false;
  }
  KS_FN void append(int claim) {     // a new claim with its first pod (scheduler.go:785)
    W::store(&key[n], 1u); W::store(&ord[n], (uint32_t)claim); if constexpr (POS) W::store(&pos[claim], (uint32_t)n);
    W::sync();
    defect = n; defect_append = true;
    n++;
  }

  // ---- Go pdqsort pieces ----
  static KS_FN int bits_len(unsigned x) { return x ? 32 - __builtin_clz(x) : 0; }

  KS_FN void insertion_sort(int a, int b) {
    for (int i = a + 1; i < b; i++)
      for (int j = i; j > a && less(j, j - 1); j--) swap(j, j - 1);
  }
  KS_FN void sift_down(int lo, int hi, int first) {
    int root = lo;
    for (;;) {
      int child = 2 * root + 1;
      if (child >= hi) return;
      if (child + 1 < hi && less(first + child, first + child + 1)) child++;
      if (!less(first + root, first + child)) return;
      swap(first + root, first + child);
      root = child;
    }
  }
  KS_FN void heap_sort(int a, int b) {
    int first = a, lo = 0, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) sift_down(i, hi, first);
    for (int i = hi - 1; i >= 0; i--) { swap(first, first + i); sift_down(lo, i, first); }
  }
  // order2/median on preloaded keys: indices are permuted, data is not touched
  static KS_FN void order2(int& a, int& b, uint32_t& ka, uint32_t& kb, int& swaps) {
    if (kb < ka) { swaps++; int t = a; a = b; b = t; uint32_t tk = ka; ka = kb; kb = tk; }
  }
  static KS_FN void median3(int a, int b, int c, uint32_t ka, uint32_t kb, uint32_t kc, int& swaps, int& out, uint32_t& kout) {
    order2(a, b, ka, kb, swaps); order2(b, c, kb, kc, swaps); order2(a, b, ka, kb, swaps);
    out = b; kout = kb;
  }
  KS_FN int choose_pivot(int a, int b, int& hint) {  // hint: 0 unknown, 1 increasing, 2 decreasing
    int l = b - a, swaps = 0;
    int i = a + l / 4 * 1, j = a + l / 4 * 2, k = a + l / 4 * 3;
    if (l >= 8) {
      uint32_t ki, kj, kk;
      if (l >= 50) {
        uint32_t s0 = key[i - 1], s1 = key[i], s2 = key[i + 1], s3 = key[j - 1], s4 = key[j], s5 = key[j + 1], s6 = key[k - 1], s7 = key[k], s8 = key[k + 1];
        int oi, oj, ok;
        median3(i - 1, i, i + 1, s0, s1, s2, swaps, oi, ki);
        median3(j - 1, j, j + 1, s3, s4, s5, swaps, oj, kj);
        median3(k - 1, k, k + 1, s6, s7, s8, swaps, ok, kk);
        i = oi; j = oj; k = ok;
      } else { ki = key[i]; kj = key[j]; kk = key[k]; }
      int oj2; uint32_t dummy;
      median3(i, j, k, ki, kj, kk, swaps, oj2, dummy);
      j = oj2;
    }
    hint = swaps == 0 ? 1 : (swaps == 12 ? 2 : 0);
    return j;
  }
  KS_FN void reverse_range(int a, int b) { int i = a, j = b - 1; while (i < j) { swap(i, j); i++; j--; } }

  // first x in [i,b) with key[x] < key[x-1]; at the top level the only possible descents are at the defect
  KS_FN int next_descent(int i, int b, bool top) {
    if (top) {
      if (defect < 0) return b;
      for (int x = defect; x <= defect + 1; ++x) if (x >= i && x >= 1 && x < b && key[x] < key[x - 1]) return x;
      return b;
    }
    const P32 kp = key;
    return ff(i, b, [kp](int x) { return kp[x] < kp[x - 1]; });
  }
  KS_FN bool partial_insertion_sort(int a, int b, bool top) {
    int i = a + 1;
    for (int step = 0; step < 5; step++) {
      i = next_descent(i, b, top);
      if (i == b) return true;
      if (b - a < 50) return false;
      if (top) {
        // Single known defect: Go's swap(i,i-1) + the two shift loops amount to ONE rotation of the touched claim to
        // its stable place (see the derivation in DESIGN.md §4); do it with one search + one rotate.
        const P32 kq = key;
        if (defect_append) {           // i == n-1: the new claim moves left behind the last claim with <= its count
          uint32_t mv = key[i];
          int t = fl(0, i, [kq, mv](int x) { return !(mv < kq[x]); });
          rotate_right(t + 1, i);
        } else {                       // i == p+1: the incremented claim at p moves right past the claims with a smaller count
          uint32_t mv = key[i - 1];
          int e = ff(i, b, [kq, mv](int x) { return !(kq[x] < mv); });
          rotate_left(i - 1, e - 1);
        }
        defect = -1;
        return true;
      }
      swap(i, i - 1);
      if (i - a >= 2) {  // shift the smaller one to the left (Go uses the absolute bound j >= 1)
        uint32_t mv = key[i - 1];
        const P32 kp = key;
        int t = fl(0, i - 1, [kp, mv](int x) { return !(mv < kp[x]); });
        rotate_right(t + 1, i - 1);
      }
      if (b - i >= 2) {  // shift the greater one to the right
        uint32_t mv = key[i];
        const P32 kp = key;
        int e = ff(i + 1, b, [kp, mv](int x) { return !(kp[x] < mv); });
        rotate_left(i, e - 1);
      }
      if (top) defect = -1;  // the single defect is repaired: the rest of the array is known sorted
    }
    return false;
  }
  KS_FN void break_patterns(int a, int b) {
    int length = b - a;
    if (length >= 8) {
      uint64_t r = (uint64_t)length;
      unsigned modulus = 1u << bits_len((unsigned)length);
      int idx = a + (length / 4) * 2 - 1;
      for (int t = 0; t < 3; t++) {
        r ^= r << 13; r ^= r >> 7; r ^= r << 17;
        int other = (int)((unsigned)r & (modulus - 1));
        if (other >= length) other -= length;
        swap(idx - 1 + t, a + other);
      }
    }
  }
  KS_FN int partition_equal(int a, int b, int pivot) {
    swap(a, pivot);
    uint32_t pv = key[a];
    const P32 kp = key;
    int i = a + 1, j = b - 1;
    for (;;) {
      i = ff(i, j + 1, [kp, pv](int x) { return pv < kp[x]; });
      j = fl(i, j + 1, [kp, pv](int x) { return !(pv < kp[x]); });
      if (i > j) break;
      swap(i, j); i++; j--;
    }
    return i;
  }
  KS_FN int partition(int a, int b, int pivot, bool& already) {
    swap(a, pivot);
    uint32_t pv = key[a];
    const P32 kp = key;
    int i = a + 1, j = b - 1;
    i = ff(i, j + 1, [kp, pv](int x) { return !(kp[x] < pv); });
    j = fl(i, j + 1, [kp, pv](int x) { return kp[x] < pv; });
    if (i > j) { swap(j, a); already = true; return j; }
    swap(i, j); i++; j--;
    for (;;) {
      i = ff(i, j + 1, [kp, pv](int x) { return !(kp[x] < pv); });
      j = fl(i, j + 1, [kp, pv](int x) { return kp[x] < pv; });
      if (i > j) break;
      swap(i, j); i++; j--;
    }
    swap(j, a);
    already = false;
    return j;
  }

  struct Frame { int a, b, limit; bool was_balanced, was_partitioned; };

  // pdqsort_func with the recursion turned into an explicit stack (the recursive call always takes the smaller
  // side, so the depth is bounded by log2 n).
  KS_FN void pdqsort(int a0, int b0, int limit0) {
    Frame stack[40];
    int sp = 0;
    Frame cur{a0, b0, limit0, true, true};
    bool top = true;  // still the outermost call, nothing swapped yet
    for (;;) {
      bool done = false;
      for (;;) {
        int a = cur.a, b = cur.b;
        int length = b - a;
        if (length <= 12) { insertion_sort(a, b); done = true; break; }
        if (cur.limit == 0) { heap_sort(a, b); done = true; break; }
        if (!cur.was_balanced) { break_patterns(a, b); cur.limit--; }
        int hint;
        int pivot = choose_pivot(a, b, hint);
        if (hint == 2) { reverse_range(a, b); pivot = (b - 1) - (pivot - a); hint = 1; top = false; }
        if (cur.was_balanced && cur.was_partitioned && hint == 1) {
          if (partial_insertion_sort(a, b, top)) { done = true; break; }
        }
        if (top) slow_sorts++;  // the outermost call leaves the single-defect fast path: full pdqsort from here on
        top = false;
        if (a > 0 && !less(a - 1, pivot)) { cur.a = partition_equal(a, b, pivot); continue; }
        bool already;
        int mid = partition(a, b, pivot, already);
        cur.was_partitioned = already;
        int left_len = mid - a, right_len = b - mid;
        int balance_threshold = length / 8;
        Frame child;
        if (left_len < right_len) {
          cur.was_balanced = left_len >= balance_threshold;
          child = Frame{a, mid, cur.limit, true, true};
          cur.a = mid + 1;
        } else {
          cur.was_balanced = right_len >= balance_threshold;
          child = Frame{mid + 1, b, cur.limit, true, true};
          cur.b = mid;
        }
        stack[sp++] = cur;  // the parent continues after the child has run to completion
        cur = child;
      }
      (void)done;
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }

  // sort.Slice on the current array
  KS_FN void sort() {
    if (defect < 0 || n <= 1) { defect = -1; return; }  // sorted input: pdqsort performs no swap
    if (n <= 12) {
      // insertionSort_func is a stable sort; with a single defect that is one stable move
      if (defect_append) {
        uint32_t mv = key[n - 1];
        const P32 kp = key;
        int t = fl(0, n - 1, [kp, mv](int x) { return !(mv < kp[x]); });
        rotate_right(t + 1, n - 1);
      } else {
        int p = defect;
        uint32_t mv = key[p];
        const P32 kp = key;
        int e = ff(p + 1, n, [kp, mv](int x) { return !(kp[x] < mv); });
        rotate_left(p, e - 1, true);   // everything between carries the claim's old count
      }
      defect = -1;
      return;
    }
    if (n >= 50) {
      // The outermost pdqsort call on "sorted except position p": choosePivot samples the keys around n/4, n/2 and 3n/4
      // (zsortfunc.go choosePivot_func); unless p is one of those nine positions every sampled comparison sees sorted
      // data, so swaps == 0, the hint is "increasing" and the call goes straight to partialInsertionSort, which repairs
      // the single defect with one stable move (see partial_insertion_sort). Skip the sampling in that case.
      const int q = n / 4, p = defect;
      const bool sampled = (p >= q - 1 && p <= q + 1) || (p >= 2 * q - 1 && p <= 2 * q + 1) || (p >= 3 * q - 1 && p <= 3 * q + 1);
      if (!sampled) {
        const P32 kq = key;
        // LDS-resident order: a vector search (usually one round); HBM-resident order: runs of equally full claims can be
        // tens of thousands long, binary search instead
        if (defect_append) {
          const int i = n - 1;
          if (i >= 1 && key[i] < key[i - 1]) {
            const uint32_t mv = key[i];
            const int t = kPerLane > 1 ? upper_last_sorted(0, i, mv) : fl(0, i, [kq, mv](int x) { return !(mv < kq[x]); });   // [0, i) is sorted
            rotate_right(t + 1, i);
          }
        } else if (p + 1 < n && key[p + 1] < key[p]) {
          const uint32_t mv = key[p];
          const int e = kPerLane > 1 ? lower_bound_sorted(p + 1, n, mv) : ff(p + 1, n, [kq, mv](int x) { return !(kq[x] < mv); });   // [p+1, n) is sorted
          rotate_left(p, e - 1, true);   // everything between carries the claim's old count
        }
        defect = -1;
        return;
      }
    }
    pdqsort(0, n, bits_len((unsigned)n));
    defect = -1;
  }
};

}  // namespace ks
