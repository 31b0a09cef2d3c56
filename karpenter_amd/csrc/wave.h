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
#define KS_LDS __attribute__((address_space(3)))   // pointers known to point into the CU's LDS: ds_* instead of flat_*
#define KS_GLOBAL __attribute__((address_space(1)))   // pointers known to point into HBM: global_* instead of flat_* (a flat access also counts against lgkmcnt, so every LDS wait behind it waits for HBM)
#else
#define KS_DEVICE 0   // g++ (host flattener, test-only emulation): a wave is a loop over 64 lanes
#define KS_FN inline
#define KS_DEV inline
#define KS_LDS
#define KS_GLOBAL
#endif

namespace ks {

#if KS_DEVICE

struct Wave {
  KS_DEV static int lane() { return (int)(threadIdx.x & 63); }
  // A value every lane holds alike, moved to scalar registers: cross-lane shuffles leave the compiler believing the result
  // is divergent, and a branch on it (and every value merged behind that branch) would be compiled as divergent code.
  KS_DEV static uint64_t uniform(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
  }
  // Orders this wave's LDS/global accesses; the wave executes in lockstep so this is a compiler + counter fence only.
  KS_DEV static void sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // Compiler-only ordering of this wavefront's memory accesses: nothing is emitted and nothing is waited for. Enough between a
  // store and a later load of the SAME memory (LDS, or one wavefront's plain global accesses): the hardware keeps a wavefront's
  // accesses to one memory in order.
  KS_DEV static void order() { asm volatile("" ::: "memory"); }
  KS_DEV static void sched_fence() { __builtin_amdgcn_sched_barrier(0); }   // the instruction scheduler moves nothing across this point
  // Between a store to HBM and another lane's load of the same address in the cursor engine's fast loop: the fence (the stores are
  // waited for). -DKS_FAST_NO_HBM_FENCE (measurement builds only, scripts/gpu_r5_f.sh) makes it a compiler-only barrier, relying on
  // the memory pipeline keeping one wavefront's accesses to an address in order.
#ifdef KS_FAST_NO_HBM_FENCE
  KS_DEV static void hbm_sync() { order(); }
#else
  KS_DEV static void hbm_sync() { sync(); }
#endif
  template <class F>
  KS_DEV static uint64_t ballot(F f) { return __builtin_amdgcn_ballot_w64((bool)f(lane())); }   // (the builtin on a bool: the compare itself writes the mask; __ballot(int) materialises 0 / 1 in a VGPR and compares again)
  // four ballots from ONE evaluation of f(lane) (bits 0..3 of its result): the loads behind the predicates happen once
  template <class F>
  KS_DEV static void ballot4(F f, uint64_t& m0, uint64_t& m1, uint64_t& m2, uint64_t& m3) {
    const int v = f(lane());
    m0 = __builtin_amdgcn_ballot_w64((v & 1) != 0); m1 = __builtin_amdgcn_ballot_w64((v & 2) != 0); m2 = __builtin_amdgcn_ballot_w64((v & 4) != 0); m3 = __builtin_amdgcn_ballot_w64((v & 8) != 0);
  }
  template <class F>
  KS_DEV static void ballot2(F f, uint64_t& m0, uint64_t& m1) {
    const int v = f(lane());
    m0 = __builtin_amdgcn_ballot_w64((v & 1) != 0); m1 = __builtin_amdgcn_ballot_w64((v & 2) != 0);
  }
  // g(j, ballot(f(lane, j))) for j < n <= 8: all predicates (and their loads) are evaluated before the first ballot, so
  // eight words cost one LDS latency instead of eight. No arrays: everything stays in registers.
  template <class F, class G>
  KS_DEV static void ballots8(int n, F f, G g) {
    const int l = lane();
    int p0 = 0 < n ? (int)f(l, 0) : 0, p1 = 1 < n ? (int)f(l, 1) : 0, p2 = 2 < n ? (int)f(l, 2) : 0, p3 = 3 < n ? (int)f(l, 3) : 0;
    int p4 = 4 < n ? (int)f(l, 4) : 0, p5 = 5 < n ? (int)f(l, 5) : 0, p6 = 6 < n ? (int)f(l, 6) : 0, p7 = 7 < n ? (int)f(l, 7) : 0;
    if (0 < n) g(0, (uint64_t)__ballot(p0)); if (1 < n) g(1, (uint64_t)__ballot(p1)); if (2 < n) g(2, (uint64_t)__ballot(p2)); if (3 < n) g(3, (uint64_t)__ballot(p3));
    if (4 < n) g(4, (uint64_t)__ballot(p4)); if (5 < n) g(5, (uint64_t)__ballot(p5)); if (6 < n) g(6, (uint64_t)__ballot(p6)); if (7 < n) g(7, (uint64_t)__ballot(p7));
  }
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
      uint64_t m = __builtin_amdgcn_ballot_w64(i < hi && pred(i));
      if (m) return base + __builtin_ctzll(m);
    }
    return hi;
  }
  // largest i in [lo,hi) with pred(i), else lo-1
  template <class F>
  KS_DEV static int find_last(int lo, int hi, F pred) {
    for (int top = hi; top > lo; top -= 64) {
      int i = top - 64 + lane();
      uint64_t m = __builtin_amdgcn_ballot_w64(i >= lo && pred(i));
      if (m) return top - 64 + (63 - __builtin_clzll(m));
    }
    return lo - 1;
  }
  // the same searches over HBM-resident arrays: eight rounds of loads in flight before the first ballot (a round trip to L2 per
  // 64 elements otherwise: a scan over 27,000 in-flight claims is 420 dependent round trips)
  template <class F>
  KS_DEV static int find_first8(int lo, int hi, F pred) {
    for (int base = lo; base < hi; base += 512) {
      int r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = base + j * 64 + lane(); r[j] = (i < hi && pred(i)) ? 1 : 0; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const uint64_t m = __ballot(r[j]); if (m) return base + j * 64 + __builtin_ctzll(m); }
    }
    return hi;
  }
  template <class F>
  KS_DEV static int find_last8(int lo, int hi, F pred) {
    for (int top = hi; top > lo; top -= 512) {
      int r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = top - (j + 1) * 64 + lane(); r[j] = (i >= lo && pred(i)) ? 1 : 0; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const uint64_t m = __ballot(r[j]); if (m) return top - (j + 1) * 64 + (63 - __builtin_clzll(m)); }
    }
    return lo - 1;
  }
  // dst[i] = src[i] for i in [0,n): eight loads in flight per lane
  template <class D, class S>
  KS_DEV static void copy8(D dst, S src, int n) {
    for (int base = 0; base < n; base += 512) {
      decltype(src[0] + 0) v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = base + j * 64 + lane(); if (i < n) v[j] = src[i]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int i = base + j * 64 + lane(); if (i < n) dst[i] = (decltype(dst[0] + 0))v[j]; }
    }
    sync();
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
    return uniform(v);
  }
  template <class F>
  KS_DEV static int64_t reduce_max_i64(int n, F f) {
    int64_t v = INT64_MIN;
    for (int i = lane(); i < n; i += 64) { int64_t x = f(i); v = x > v ? x : v; }
    for (int off = 32; off > 0; off >>= 1) {
      int64_t o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
    return (int64_t)uniform((uint64_t)v);
  }
  // max over the 64 lanes of f(lane)
  template <class F>
  KS_DEV static int64_t lanes_max_i64(F f) {
    int64_t v = f(lane());
    for (int off = 32; off > 0; off >>= 1) {
      int64_t o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
    return (int64_t)uniform((uint64_t)v);
  }
  template <class F>
  KS_DEV static uint64_t reduce_or(int n, F f) {
    uint64_t v = 0;
    for (int i = lane(); i < n; i += 64) v |= f(i);
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return uniform(v);
  }
  // scalar store: one lane writes, every lane may read it back afterwards (same wave, program order)
  template <class T, class V>
  KS_DEV static void store(T* p, V v) {
    if (lane() == 0) *p = (T)v;
  }
  template <class T, class V>
  KS_DEV static void store(KS_LDS T* p, V v) {
    if (lane() == 0) *p = (T)v;
  }
  KS_DEV static bool leader() { return lane() == 0; }
  // f(lane) on every lane (per-lane register state; no result)
  template <class F>
  KS_DEV static void each(F f) { f(lane()); }
  // A flag another agent (the host, over PCIe) may set while the kernel runs: a system-scope atomic load, so that it is
  // not served from the scalar or vector caches forever.
  KS_DEV static int poll_flag(const volatile int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
  // Shader clock for the per-phase counters of a profiling build (-DKSOLVE_PHASE_TIMERS, scripts/gpu_quick.sh). The
  // product build compiles the timers out: 24 live 64-bit accumulators and an s_memtime + s_waitcnt per phase boundary
  // cost a single wavefront ~10% and a lot of register pressure.
#ifdef KSOLVE_PHASE_TIMERS
  KS_DEV static unsigned long long clock() { return __builtin_readcyclecounter(); }
#else
  KS_DEV static unsigned long long clock() { return 0ull; }
#endif
  // min over the 64 lanes of a u32 with DPP row operations (no LDS traffic): quad swaps, row rotates, row broadcasts;
  // the full result lands in lane 63.
  KS_DEV static uint32_t min_u32(uint32_t v) {
    int x = (int)v;
#define KS_DPP_MIN(ctrl, row_mask) { int o = __builtin_amdgcn_update_dpp(-1, x, ctrl, row_mask, 0xf, false); x = ((uint32_t)o < (uint32_t)x) ? o : x; }
    KS_DPP_MIN(0xB1, 0xf)    // quad_perm [1,0,3,2]
    KS_DPP_MIN(0x4E, 0xf)    // quad_perm [2,3,0,1]
    KS_DPP_MIN(0x124, 0xf)   // row_ror:4
    KS_DPP_MIN(0x128, 0xf)   // row_ror:8
    KS_DPP_MIN(0x142, 0xa)   // row_bcast:15 -> rows 1,3
    KS_DPP_MIN(0x143, 0xc)   // row_bcast:31 -> rows 2,3
#undef KS_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
  }
  // smallest f(lane) over the lanes with valid(lane), as (value, lane); value = ~0u when none
  template <class F>
  KS_DEV static uint32_t argmin_u32(F f, int* lane_out) {
    uint32_t mine = f(lane());
    uint32_t m = min_u32(mine);
    uint64_t who = __builtin_amdgcn_ballot_w64(mine == m);
    *lane_out = m == 0xFFFFFFFFu ? -1 : (int)__builtin_ctzll(who);
    return m;
  }
};

#else  // host emulation (tests only)

// Lanes run one after the other here and in lockstep on the device, so a lambda that read what another lane of the same
// call wrote would behave differently in the two. -DKS_EMU_REVERSE_LANES runs the lanes of every wave-wide call in the
// opposite order: the fuzzers must give the same answers in both orders (tests/test_device_algorithm.py).
#ifdef KS_EMU_REVERSE_LANES
#define KS_LANES(l) for (int l = 63; l >= 0; --l)
#else
#define KS_LANES(l) for (int l = 0; l < 64; ++l)
#endif

struct Wave {
  static int lane() { return 0; }
  static uint64_t uniform(uint64_t v) { return v; }
  static void sync() {}
  static void order() {}
  static void sched_fence() {}
  static void hbm_sync() {}
  template <class F>
  static uint64_t ballot(F f) {
    uint64_t m = 0;
    KS_LANES(l) if (f(l)) m |= 1ull << l;
    return m;
  }
  template <class F>
  static void ballot4(F f, uint64_t& m0, uint64_t& m1, uint64_t& m2, uint64_t& m3) {
    m0 = m1 = m2 = m3 = 0;
    KS_LANES(l) { const int v = f(l); if (v & 1) m0 |= 1ull << l; if (v & 2) m1 |= 1ull << l; if (v & 4) m2 |= 1ull << l; if (v & 8) m3 |= 1ull << l; }
  }
  template <class F>
  static void ballot2(F f, uint64_t& m0, uint64_t& m1) {
    m0 = m1 = 0;
    KS_LANES(l) { const int v = f(l); if (v & 1) m0 |= 1ull << l; if (v & 2) m1 |= 1ull << l; }
  }
  template <class F, class G>
  static void ballots8(int n, F f, G g) {
    for (int j = 0; j < n; ++j) { uint64_t m = 0; KS_LANES(l) if (f(l, j)) m |= 1ull << l; g(j, m); }
  }
  // the device gives item i to lane i % 64: items of one round of 64 run together, rounds one after the other
  template <class F>
  static void for_n(int n, F f) {
    for (int base = 0; base < n; base += 64) KS_LANES(l) if (base + l < n) f(base + l);
  }
  template <class F>
  static int find_first(int lo, int hi, F pred) { for (int i = lo; i < hi; ++i) if (pred(i)) return i; return hi; }
  template <class F>
  static int find_last(int lo, int hi, F pred) { for (int i = hi - 1; i >= lo; --i) if (pred(i)) return i; return lo - 1; }
  template <class F>
  static uint64_t reduce_min(int n, F f) { uint64_t v = ~0ull; for_n(n, [&](int i) { uint64_t x = f(i); if (x < v) v = x; }); return v; }
  template <class F>
  static int find_first8(int lo, int hi, F pred) { return find_first(lo, hi, pred); }
  template <class F>
  static int find_last8(int lo, int hi, F pred) { return find_last(lo, hi, pred); }
  template <class D, class S>
  static void copy8(D dst, S src, int n) { for (int i = 0; i < n; ++i) dst[i] = src[i]; }
  template <class F>
  static int64_t reduce_max_i64(int n, F f) { int64_t v = INT64_MIN; for_n(n, [&](int i) { int64_t x = f(i); if (x > v) v = x; }); return v; }
  template <class F>
  static int64_t lanes_max_i64(F f) { int64_t v = INT64_MIN; KS_LANES(l) { int64_t x = f(l); if (x > v) v = x; } return v; }
  template <class F>
  static uint64_t reduce_or(int n, F f) { uint64_t v = 0; for_n(n, [&](int i) { v |= f(i); }); return v; }
  template <class T, class V>
  static void store(T* p, V v) { *p = (T)v; }
  static bool leader() { return true; }
  template <class F>
  static void each(F f) { KS_LANES(l) f(l); }
#if defined(KSOLVE_PHASE_TIMERS) && defined(__x86_64__)
  static unsigned long long clock() { return __builtin_ia32_rdtsc(); }   // host profile of the emulation (tests/tools): where the work is, not the latency
#else
  static unsigned long long clock() { return 0; }
#endif
  static int poll_flag(const volatile int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
  template <class F>
  static uint32_t argmin_u32(F f, int* lane_out) {   // ties go to the lowest lane, whatever the evaluation order
    uint32_t m = 0xFFFFFFFFu; int who = -1;
    KS_LANES(l) { uint32_t x = f(l); if (x < m || (x == m && who >= 0 && l < who)) { m = x; who = l; } }
    *lane_out = who;
    return m;
  }
};

#endif

// One u64 per lane, written from wave-uniform values (v_writelane: no LDS, no exec-mask juggling).
struct LaneVec64 {
#if KS_DEVICE
  uint64_t v = 0;
  KS_DEV void set(int lane, uint64_t x) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    // one SGPR operand per VOP3 (constant bus): the lane select goes through m0, which is put back afterwards (m0 is a
    // reserved register: the compiler does not honour it in a clobber list)
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
                 : "+v"(lo), "+v"(hi), "=&s"(m0_saved) : "s"(xl), "s"(lane), "s"(xh));
    v = (uint64_t)lo | ((uint64_t)hi << 32);
  }
  KS_DEV uint64_t get(int) const { return v; }
#else
  uint64_t v[64] = {};
  void set(int lane, uint64_t x) { v[lane] = x; }
  uint64_t get(int l) const { return v[l]; }
#endif
};

// One value per lane that lives in a vector register on the device (an array of 64 in the host emulation);
// bcast(lane) makes one lane's value wave-uniform (v_readlane, no LDS round trip).
template <class T>
struct LaneVar {
#if KS_DEVICE
  T v;
  KS_DEV T& at(int) { return v; }
  KS_DEV T v_of(int) const { return v; }   // this lane's value (const access)
  KS_DEV void set(int lane, T x) {   // a wave-uniform value into one lane (v_writelane)
    static_assert(sizeof(T) == 4, "LaneVar::set: 32-bit values");
    uint32_t w = __builtin_bit_cast(uint32_t, v), m0_saved;
    const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_bit_cast(uint32_t, x));   // in an SGPR, never a literal
    lane = __builtin_amdgcn_readfirstlane(lane);
    // one SGPR operand per VOP3 (constant bus): the lane select goes through m0, which is put back afterwards
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(w), "=&s"(m0_saved) : "s"(xs), "s"(lane));
    v = __builtin_bit_cast(T, w);
  }
  // this lane reads lane `src`'s value (ds_bpermute_b32: the LDS crossbar, no LDS memory); `src` may differ per lane
  KS_DEV T shuffle(int, int src) const {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "LaneVar::shuffle: 32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
      const int x = __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v));
      return __builtin_bit_cast(T, x);
    } else {
      const uint64_t u = __builtin_bit_cast(uint64_t, v);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)u);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(u >> 32));
      return __builtin_bit_cast(T, (uint64_t)lo | ((uint64_t)hi << 32));
    }
  }
  KS_DEV T bcast(int lane) const {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "LaneVar: 32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
      const int x = __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane);
      return __builtin_bit_cast(T, x);
    } else {
      const uint64_t u = __builtin_bit_cast(uint64_t, v);
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
      return __builtin_bit_cast(T, (uint64_t)lo | ((uint64_t)hi << 32));
    }
  }
#else
  T v[64];
  T& at(int l) { return v[l]; }
  T v_of(int l) const { return v[l]; }
  void set(int lane, T x) { v[lane] = x; }
  T shuffle(int, int src) const { return v[src & 63]; }   // callers never read a lane the same wave-wide call writes
  T bcast(int lane) const { return v[lane]; }
#endif
};

KS_FN int popc64(uint64_t x) { return __builtin_popcountll(x); }
KS_FN int ctz64(uint64_t x) { return __builtin_ctzll(x); }

}  // namespace ks
