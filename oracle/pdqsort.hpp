// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Go standard library sort.Slice / sort.SliceStable (go1.26, src/sort/slice.go + zsortfunc.go), restated.
// The Go toolchain and its sources are NOT on this machine and not under /root/reference (the reference depends on
// the stdlib implicitly, go.mod:3 "go 1.26.6"), so this is restated from the published algorithm
// (pattern-defeating quicksort, Orson Peters; Go port zsortfunc.go). PARITY UNPINNED: no reference test pins the
// order sort.Slice leaves equal elements in; it matters at scheduler.go:598 (claims sorted by len(Pods), ties
// decide which claim a pod lands in), requirements.go:102 and types.go:338.
// less(i,j) and swap(i,j) act on positions of the caller's slice exactly like Go's reflect-swapper closures.
#pragma once
#include <cstdint>
#include <functional>

namespace oracle {

// LessFn / SwapFn: the two closures Go's sort.Slice builds (less(i, j) on positions, the reflect swapper)
template <class LessFn, class SwapFn>
struct GoSort {
  LessFn less;
  SwapFn swap;
  GoSort(LessFn l, SwapFn s) : less(l), swap(s) {}

  enum Hint { unknownHint = 0, increasingHint, decreasingHint };

  static int bits_len(unsigned long long x) { int n = 0; while (x) { ++n; x >>= 1; } return n; }

  // insertionSort_func
  void insertion_sort(int a, int b) {
    for (int i = a + 1; i < b; i++)
      for (int j = i; j > a && less(j, j - 1); j--) swap(j, j - 1);
  }
  // siftDown_func
  void sift_down(int lo, int hi, int first) {
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
  // heapSort_func
  void heap_sort(int a, int b) {
    int first = a, lo = 0, hi = b - a;
    for (int i = (hi - 1) / 2; i >= 0; i--) sift_down(i, hi, first);
    for (int i = hi - 1; i >= 0; i--) { swap(first, first + i); sift_down(lo, i, first); }
  }
  // order2_func / median_func / medianAdjacent_func
  void order2(int& a, int& b, int& swaps) { if (less(b, a)) { swaps++; int t = a; a = b; b = t; } }
  int median(int a, int b, int c, int& swaps) { order2(a, b, swaps); order2(b, c, swaps); order2(a, b, swaps); return b; }
  int median_adjacent(int a, int& swaps) { return median(a - 1, a, a + 1, swaps); }
  // choosePivot_func
  int choose_pivot(int a, int b, Hint& hint) {
    const int shortestNinther = 50, maxSwaps = 4 * 3;
    int l = b - a, swaps = 0;
    int i = a + l / 4 * 1, j = a + l / 4 * 2, k = a + l / 4 * 3;
    if (l >= 8) {
      if (l >= shortestNinther) { i = median_adjacent(i, swaps); j = median_adjacent(j, swaps); k = median_adjacent(k, swaps); }
      j = median(i, j, k, swaps);
    }
    if (swaps == 0) hint = increasingHint;
    else if (swaps == maxSwaps) hint = decreasingHint;
    else hint = unknownHint;
    return j;
  }
  void reverse_range(int a, int b) { int i = a, j = b - 1; while (i < j) { swap(i, j); i++; j--; } }
  // partialInsertionSort_func
  bool partial_insertion_sort(int a, int b) {
    const int maxSteps = 5, shortestShifting = 50;
    int i = a + 1;
    for (int j = 0; j < maxSteps; j++) {
      while (i < b && !less(i, i - 1)) i++;
      if (i == b) return true;
      if (b - a < shortestShifting) return false;
      swap(i, i - 1);
      if (i - a >= 2) for (int jj = i - 1; jj >= 1; jj--) { if (!less(jj, jj - 1)) break; swap(jj, jj - 1); }
      if (b - i >= 2) for (int jj = i + 1; jj < b; jj++) { if (!less(jj, jj - 1)) break; swap(jj, jj - 1); }
    }
    return false;
  }
  // breakPatterns_func (xorshift seeded with the length; nextPowerOfTwo = 1 << bits.Len(length))
  void break_patterns(int a, int b) {
    int length = b - a;
    if (length >= 8) {
      uint64_t r = (uint64_t)length;
      unsigned modulus = 1u << bits_len((unsigned)length);
      int idx = a + (length / 4) * 2 - 1;
      for (int i = 0; i < 3; i++) {
        r ^= r << 13; r ^= r >> 7; r ^= r << 17;
        int other = (int)((unsigned)r & (modulus - 1));
        if (other >= length) other -= length;
        swap(idx - 1 + i, a + other);
      }
    }
  }
  // partitionEqual_func
  int partition_equal(int a, int b, int pivot) {
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
  // partition_func
  int partition(int a, int b, int pivot, bool& already) {
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
  // pdqsort_func
  void pdqsort(int a, int b, int limit) {
    const int maxInsertion = 12;
    bool wasBalanced = true, wasPartitioned = true;
    for (;;) {
      int length = b - a;
      if (length <= maxInsertion) { insertion_sort(a, b); return; }
      if (limit == 0) { heap_sort(a, b); return; }
      if (!wasBalanced) { break_patterns(a, b); limit--; }
      Hint hint;
      int pivot = choose_pivot(a, b, hint);
      if (hint == decreasingHint) {
        reverse_range(a, b);
        pivot = (b - 1) - (pivot - a);
        hint = increasingHint;
      }
      if (wasBalanced && wasPartitioned && hint == increasingHint) {
        if (partial_insertion_sort(a, b)) return;
      }
      if (a > 0 && !less(a - 1, pivot)) { int mid = partition_equal(a, b, pivot); a = mid; continue; }
      bool already;
      int mid = partition(a, b, pivot, already);
      wasPartitioned = already;
      int leftLen = mid - a, rightLen = b - mid;
      int balanceThreshold = length / 8;
      if (leftLen < rightLen) { wasBalanced = leftLen >= balanceThreshold; pdqsort(a, mid, limit); a = mid + 1; }
      else { wasBalanced = rightLen >= balanceThreshold; pdqsort(mid + 1, b, limit); b = mid; }
    }
  }
  // sort.Slice: pdqsort_func(lessSwap, 0, n, bits.Len(uint(n)))
  void sort_slice(int n) { pdqsort(0, n, bits_len((unsigned long long)n)); }
};

// sort.Slice over a std::vector with a value comparator.
template <class T, class Less>
void go_sort_slice(std::vector<T>& v, Less lt) {
  auto less = [&](int i, int j) { return lt(v[i], v[j]); };
  auto swap = [&](int i, int j) { std::swap(v[i], v[j]); };
  GoSort<decltype(less), decltype(swap)> s(less, swap);
  s.sort_slice((int)v.size());
}

}  // namespace oracle
