// wave.h — the 64-lane wavefront as a programming model.
//
// The pack engine runs ONE wavefront per independent scheduling problem: control flow is wave-uniform (it is the
// reference's serial Solve() loop), and the lanes are used as a 64-wide vector unit for the three data-parallel
// pieces of each step: per-instance-type fit tests (one lane per instance type, __ballot -> one u64 mask word),
// first-fit searches over ordered candidate lists (ballot + ffs, "lowest index wins" like scheduler.go:639), and
// bulk moves. All of it is wave-synchronous: no __syncthreads, no atomics.
//
// The same source also compiles for the host (KSOLVE_HOST_EMULATION, used only by tests/host_engine to fuzz the
// device algorithm against the oracle on machines without a GPU): there a "wave" is a plain loop over 64 lanes.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(KSOLVE_HOST_EMULATION)
#define KS_DEVICE 1   // compiled by hipcc: the Wave below is the real 64-lane wavefront
#define KS_FN __host__ __device__ __forceinline__
#define KS_DEV __device__ __forceinline__
#else
#define KS_DEVICE 0   // g++ (host flattener, test-only emulation): a wave is a loop over 64 lanes
#define KS_FN inline
#define KS_DEV inline
#endif

namespace ks {

#if KS_DEVICE

struct Wave {
  KS_DEV static int lane() { return (int)(threadIdx.x & 63); }
  // Orders this wave's LDS/global accesses; the wave executes in lockstep so this is a compiler + counter fence only.
  KS_DEV static void sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  template <class F>
  KS_DEV static uint64_t ballot(F f) { return __ballot(f(lane()) ? 1 : 0); }
  // f(i) for i in [0,n), striped over lanes
  template <class F>
  KS_DEV static void for_n(int n, F f) {
    for (int i = lane(); i < n; i += 64) f(i);
    sync();
  }
  // smallest i in [lo,hi) with pred(i), else hi
  template <class F>
  KS_DEV static int find_first(int lo, int hi, F pred) {
    for (int base = lo; base < hi; base += 64) {
      int i = base + lane();
      uint64_t m = __ballot((i < hi && pred(i)) ? 1 : 0);
      if (m) return base + __builtin_ctzll(m);
    }
    return hi;
  }
  // largest i in [lo,hi) with pred(i), else lo-1
  template <class F>
  KS_DEV static int find_last(int lo, int hi, F pred) {
    for (int top = hi; top > lo; top -= 64) {
      int i = top - 64 + lane();
      uint64_t m = __ballot((i >= lo && pred(i)) ? 1 : 0);
      if (m) return top - 64 + (63 - __builtin_clzll(m));
    }
    return lo - 1;
  }
  // min over i in [0,n) of f(i) (u64); identity = ~0
  template <class F>
  KS_DEV static uint64_t reduce_min(int n, F f) {
    uint64_t v = ~0ull;
    for (int i = lane(); i < n; i += 64) { uint64_t x = f(i); v = x < v ? x : v; }
    for (int off = 32; off > 0; off >>= 1) {
      uint64_t o = __shfl_xor(v, off, 64);
      v = o < v ? o : v;
    }
    return v;
  }
  template <class F>
  KS_DEV static int64_t reduce_max_i64(int n, F f) {
    int64_t v = INT64_MIN;
    for (int i = lane(); i < n; i += 64) { int64_t x = f(i); v = x > v ? x : v; }
    for (int off = 32; off > 0; off >>= 1) {
      int64_t o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
    return v;
  }
  template <class F>
  KS_DEV static uint64_t reduce_or(int n, F f) {
    uint64_t v = 0;
    for (int i = lane(); i < n; i += 64) v |= f(i);
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
  }
  // scalar store: one lane writes, every lane may read it back afterwards (same wave, program order)
  template <class T>
  KS_DEV static void store(T* p, T v) {
    if (lane() == 0) *p = v;
  }
  KS_DEV static bool leader() { return lane() == 0; }
};

#else  // host emulation (tests only)

struct Wave {
  static int lane() { return 0; }
  static void sync() {}
  template <class F>
  static uint64_t ballot(F f) {
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) if (f(l)) m |= 1ull << l;
    return m;
  }
  template <class F>
  static void for_n(int n, F f) { for (int i = 0; i < n; ++i) f(i); }
  template <class F>
  static int find_first(int lo, int hi, F pred) { for (int i = lo; i < hi; ++i) if (pred(i)) return i; return hi; }
  template <class F>
  static int find_last(int lo, int hi, F pred) { for (int i = hi - 1; i >= lo; --i) if (pred(i)) return i; return lo - 1; }
  template <class F>
  static uint64_t reduce_min(int n, F f) { uint64_t v = ~0ull; for (int i = 0; i < n; ++i) { uint64_t x = f(i); if (x < v) v = x; } return v; }
  template <class F>
  static int64_t reduce_max_i64(int n, F f) { int64_t v = INT64_MIN; for (int i = 0; i < n; ++i) { int64_t x = f(i); if (x > v) v = x; } return v; }
  template <class F>
  static uint64_t reduce_or(int n, F f) { uint64_t v = 0; for (int i = 0; i < n; ++i) v |= f(i); return v; }
  template <class T>
  static void store(T* p, T v) { *p = v; }
  static bool leader() { return true; }
};

#endif

KS_FN int popc64(uint64_t x) { return __builtin_popcountll(x); }
KS_FN int ctz64(uint64_t x) { return __builtin_ctzll(x); }

}  // namespace ks
