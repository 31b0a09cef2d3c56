// go_sort.h — Go's sort.Slice (go1.26 src/sort/zsortfunc.go: pattern-defeating quicksort) over an index array, for the
// places where the reference's result depends on the permutation an UNSTABLE sort leaves among equal keys:
//     InstanceTypes.OrderByPrice (pkg/cloudprovider/types.go:336-355), used by Truncate (:437-449) when
//     Results.TruncateInstanceTypes (scheduler.go:419-437) caps a NodeClaim's instance types at 600.
// One thread sorts one claim's instance types (ksolve_finalize), so this is plain sequential code usable on host and
// device; the recursion of pdqsort_func is an explicit stack (it always recurses into the smaller side: depth <= log2 n).
// The claim ordering inside the pack engine has its own wave-parallel emulation of the same algorithm (pdq_emul.h).
#pragma once
#include "wave.h"

namespace ks {

// Less(i, j): element at position i orders before the element at position j. Swap(i, j): exchange positions.
template <class Less, class Swap>
struct GoSort {
  Less less;
  Swap swap;

  static KS_FN int bits_len(unsigned x) { int n = 0; while (x) { ++n; x >>= 1; } return n; }

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
    const int first = a, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) sift_down(i, hi, first);
    for (int i = hi - 1; i >= 0; i--) { swap(first, first + i); sift_down(0, i, first); }
  }
  KS_FN void order2(int& a, int& b, int& swaps) { if (less(b, a)) { swaps++; const int t = a; a = b; b = t; } }
  KS_FN int median(int a, int b, int c, int& swaps) { order2(a, b, swaps); order2(b, c, swaps); order2(a, b, swaps); return b; }
  // hint: 0 unknown, 1 increasing, 2 decreasing
  KS_FN int choose_pivot(int a, int b, int& hint) {
    const int l = b - a;
    int swaps = 0;
    int i = a + l / 4 * 1, j = a + l / 4 * 2, k = a + l / 4 * 3;
    if (l >= 8) {
      if (l >= 50) { i = median(i - 1, i, i + 1, swaps); j = median(j - 1, j, j + 1, swaps); k = median(k - 1, k, k + 1, swaps); }
      j = median(i, j, k, swaps);
    }
    hint = swaps == 0 ? 1 : (swaps == 12 ? 2 : 0);
    return j;
  }
  KS_FN void reverse_range(int a, int b) { for (int i = a, j = b - 1; i < j; i++, j--) swap(i, j); }
  KS_FN bool partial_insertion_sort(int a, int b) {
    int i = a + 1;
    for (int step = 0; step < 5; step++) {
      while (i < b && !less(i, i - 1)) i++;
      if (i == b) return true;
      if (b - a < 50) return false;
      swap(i, i - 1);
      if (i - a >= 2) for (int j = i - 1; j >= 1; j--) { if (!less(j, j - 1)) break; swap(j, j - 1); }
      if (b - i >= 2) for (int j = i + 1; j < b; j++) { if (!less(j, j - 1)) break; swap(j, j - 1); }
    }
    return false;
  }
  KS_FN void break_patterns(int a, int b) {
    const int length = b - a;
    if (length < 8) return;
    uint64_t r = (uint64_t)length;
    const unsigned modulus = 1u << bits_len((unsigned)length);
    const int idx = a + (length / 4) * 2 - 1;
    for (int t = 0; t < 3; t++) {
      r ^= r << 13; r ^= r >> 7; r ^= r << 17;
      int other = (int)((unsigned)r & (modulus - 1));
      if (other >= length) other -= length;
      swap(idx - 1 + t, a + other);
    }
  }
  KS_FN int partition_equal(int a, int b, int pivot) {
    swap(a, pivot);
    int i = a + 1, j = b - 1;
    for (;;) {
      while (i <= j && !less(a, i)) i++;
      while (i <= j && less(a, j)) j--;
      if (i > j) break;
      swap(i, j); i++; j--;
    }
    return i;
  }
  KS_FN int partition(int a, int b, int pivot, bool& already) {
    swap(a, pivot);
    int i = a + 1, j = b - 1;
    while (i <= j && less(i, a)) i++;
    while (i <= j && !less(j, a)) j--;
    if (i > j) { swap(j, a); already = true; return j; }
    swap(i, j); i++; j--;
    for (;;) {
      while (i <= j && less(i, a)) i++;
      while (i <= j && !less(j, a)) j--;
      if (i > j) break;
      swap(i, j); i++; j--;
    }
    swap(j, a);
    already = false;
    return j;
  }
  struct Frame { int a, b, limit; bool was_balanced, was_partitioned; };
  // sort.Slice: pdqsort_func(data, 0, n, bits.Len(uint(n)))
  KS_FN void sort(int n) {
    Frame stack[40];
    int sp = 0;
    Frame cur{0, n, bits_len((unsigned)n), true, true};
    for (;;) {
      for (;;) {
        const int a = cur.a, b = cur.b, length = b - a;
        if (length <= 12) { insertion_sort(a, b); break; }
        if (cur.limit == 0) { heap_sort(a, b); break; }
        if (!cur.was_balanced) { break_patterns(a, b); cur.limit--; }
        int hint;
        int pivot = choose_pivot(a, b, hint);
        if (hint == 2) { reverse_range(a, b); pivot = (b - 1) - (pivot - a); hint = 1; }
        if (cur.was_balanced && cur.was_partitioned && hint == 1 && partial_insertion_sort(a, b)) break;
        if (a > 0 && !less(a - 1, pivot)) { cur.a = partition_equal(a, b, pivot); continue; }
        bool already;
        const int mid = partition(a, b, pivot, already);
        cur.was_partitioned = already;
        const int left_len = mid - a, right_len = b - mid, balance_threshold = length / 8;
        Frame child;
        if (left_len < right_len) { cur.was_balanced = left_len >= balance_threshold; child = Frame{a, mid, cur.limit, true, true}; cur.a = mid + 1; }
        else { cur.was_balanced = right_len >= balance_threshold; child = Frame{mid + 1, b, cur.limit, true, true}; cur.b = mid; }
        stack[sp++] = cur;   // the parent resumes once the child call has returned
        cur = child;
      }
      if (sp == 0) return;
      cur = stack[--sp];
    }
  }
};

template <class Less, class Swap>
KS_FN void go_sort_slice(int n, Less less, Swap swap) {
  GoSort<Less, Swap> s{less, swap};
  s.sort(n);
}

}  // namespace ks
