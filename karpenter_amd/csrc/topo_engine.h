// topo_engine.h — the spread engine: Scheduler.Solve() (scheduler.go:440-519) for provisioning batches whose requirement
// algebra is purely positive (what fast_engine.h solves) AND whose pods carry topology constraints of the kinds the reference's
// own benchmark mixes (scheduling_benchmark_test.go:259-455, BASELINE configs[2]): topology spread on a dictionary key ("zonal"),
// topology spread on kubernetes.io/hostname, pod affinity on a dictionary key, pod anti-affinity on kubernetes.io/hostname (with
// its inverse group). One wavefront per problem, like every pack kernel here; what it does differently from engine.h:
//
//  1. InstanceTypeOptions are a function of (requirement set, requests) for positive sets (fast_engine.h, fact 1): CanAdd's
//     filterInstanceTypesByRequirements (nodeclaim.go:541-638) is a dominance test against the cached Pareto-maximal allocatable
//     vectors of F(requirement set) — the cursor engine's cache (FastCold::create_entry), one 32-byte LDS read per lane. The masks
//     are materialised after the loop by ksolve_fast_records.
//  2. The topology domain choice stands IN FRONT of that test. A group on a dictionary key has at most sixteen domains: its
//     counters live in LDS, nextDomainTopologySpread / nextDomainAffinity (topologygroup.go:229-298, :324-388) are evaluated once
//     per pod into a candidate-domain mask and one priority key per domain (count, name rank); every lane then narrows ITS claim's
//     field of the key with an AND and a minimum. The key's field is part of the claim's packed requirement word (`vmask`).
//  3. A hostname group's domains are the claims themselves: its counter is a 4-bit field (3 bits + a guard bit, saturating at 7)
//     of ONE 64-bit word inside the claim record, so "count + self <= maxSkew" / "count == 0" for every hostname group of the pod
//     (topologygroup.go:236-249, :407-415) is one subtraction and one compare, and Record (topology.go:197-220) one addition.
//     No per-claim bitmaps (kv_claims / host_le / c_keymask), no headroom tables: a claim is 32 bytes.
//  4. addToInflightNode's "lowest index that accepts" (scheduler.go:667-686): the reference's order is by pod count (scheduler.go:598,
//     kept bit-exactly by run_order.h's rings), so pods go to the emptiest claims and the first acceptor is almost always among the
//     first 64 positions: one lane per position, ballot, first set bit. A class with a zero limit on an anti-affinity counter (every
//     member needs a claim of its own) reads the short list of claims that hold no member yet instead — usually empty: the pod goes
//     straight to addToNewNodeClaim.
//
// Whatever it does not handle (node filters, minDomains, groups created by relaxation, anti-affinity on a dictionary key, affinity
// on the hostname, several dictionary-key groups owned by one pod, an unschedulable pod, NodePool limits that exclude a type ...)
// makes it stop with status 3 before it has written a result; the host runs the general engine on the same problem. There is no
// CPU path.
#pragma once
#include "fast_engine.h"
#include "run_order.h"

namespace ks {

constexpr int kTopoMaxGroups = 128;   // topology groups of a problem this engine takes
constexpr int kTopoMaxHost = 16;      // of them on kubernetes.io/hostname (regular + inverse): one 4-bit field each
constexpr int kTopoMaxZg = 64;        // ... and on dictionary keys
constexpr int kTopoMaxDom = 16;       // domains of such a key
constexpr int kTopoTrack = 4;         // anti-affinity counters with a list of the claims that hold no member
constexpr int kTopoFreeCap = 1024;    // entries of such a list (more: the list is dropped, its classes scan the order)
constexpr uint64_t kTopoGuard = 0x8888888888888888ull, kTopoOnes = 0x1111111111111111ull;

struct TopoRec { uint64_t vmask; int32_t req[4]; uint64_t hcnt; };   // 32 B: an in-flight claim (requirement set, requests, hostname-group counters)
// a pod class's topology: limits on the hostname counters it is tested against (field = 8 | limit; 8 | 7 where it has none),
// the counters and dictionary-key groups a pod of the class is counted by, the dictionary-key group it owns
struct TopoClass { uint64_t hlim, hinc, zsel; int32_t zg; uint32_t zself; uint32_t excl; uint32_t pad; };   // 48 B
struct TopoZg {   // a group on a dictionary key (LDS)
  int32_t cnt[kTopoMaxDom];
  uint16_t rank[kTopoMaxDom];
  uint32_t dom;        // registered domains (TopologyGroup.domains)
  int32_t nonzero;     // domains with a positive count
  int32_t skew;
  uint8_t type, var, p0, p1;   // 0 spread / 1 affinity; index of its key among the variable keys (FastMisc::vkey)
};
struct TopoState {   // LDS
  TopoZg zg[kTopoMaxZg];
  uint32_t freel[kTopoTrack][kTopoFreeCap];
  int32_t n_free[kTopoTrack];
  uint32_t track_field[kTopoTrack];   // hostname counter of list t; 0xFF: none / dropped
  int16_t gmap[kTopoMaxGroups];       // group -> hostname counter (0..15) | 0x100 + dictionary-key group | -1
  uint64_t zkey[kTopoMaxDom];         // the pod's priority key per domain (~0: not a candidate)
  uint32_t blk_cls[64], blk_claim[64], blk_cnt[64];
  TopoClass blk_tc[64];
  FastSlot blk_cs[64];
};
struct TopoPlan { int total_bytes, off_run, off_state; };   // (the cursor engine's tables sit where FastWork::plan says)
struct TopoWork { TopoClass* cls; TopoRec* rec; TopoPlan plan; int enabled; };
struct TopoArgs { ProblemView pv; Workspace ws; FastWork fw; TopoWork tw; };

template <class W>
struct TopoEngine {
  FastCold<W, 2, 1> cold;   // the cursor engine's set-up, requirement-set cache and Pareto vectors
  RunOrder<W> order;        // Go's sort.Slice permutation as one ring per pod count
  const ProblemView* Pk; const Workspace* Sk; const FastWork* Fk; const TopoWork* Tk;
  KS_LDS TopoState* st;
  KS_LDS FastMisc* Mp;
  int n_zg = 0, n_host = 0, n_track = 0;
  uint64_t track_fields = 0;   // bit 4 f: counter f has a list
  unsigned long long n_ref = 0, n_tests = 0, n_windows = 0, n_listed = 0;
  int bail = 0;

  KS_DEV TopoEngine(const ProblemView* p, const Workspace* s, const FastWork* f, const TopoWork* t, char* lds) {
    Pk = p; Sk = s; Fk = f; Tk = t;
    cold.init(p, s, f, lds);
    Mp = cold.Mp;
    st = (KS_LDS TopoState*)(lds + t->plan.off_state);
    order.init((KS_LDS RunTables*)(lds + t->plan.off_run), s->o_ring, s->o_cnt, s->o_pos, s->o_key, s->o_ord, s->run_tabs, s->run_off, s->run_log, s->run_kmax);
  }

  // ---- the shape check and the topology tables; 0 = this engine solves the problem ----
  KS_COLD int setup_topo() {
    const ProblemView& P = *Pk; const TopoView& T = P.topo; const Dict& d = P.dict;
    const int G = T.n_groups;
    if (G <= 0 || G > kTopoMaxGroups || T.n_alias) return 40;
    if (!P.strict_same) return 41;   // podDomains = StrictRequirements (topology.go:230): one table when no pod has a preference
    KS_LDS TopoState& S_ = *st;
    const int nvv = cold.nv;
    n_zg = 0; n_host = 0; n_track = 0; track_fields = 0;
    for (int t = 0; t < kTopoTrack; ++t) { W::store(&S_.n_free[t], 0); W::store(&S_.track_field[t], 0xFFu); }
    const bool any_taint = W::reduce_or(P.n_templates, [&](int t) { return P.tmpl_taints[t]; }) != 0;
    for (int g = 0; g < G; ++g) {
      const bool inv = (T.inverse_mask[g >> 6] >> (g & 63)) & 1;
      if (!((T.initially_active[g >> 6] >> (g & 63)) & 1)) return 42;             // created by a relaxing pod (topology.go:162-194)
      {
        // TopologyNodeFilter.Matches (topologynodefilter.go:68-96) must be true for every claim: no taint policy to honor (or no
        // tainted template), and no node-affinity terms — or one that requires nothing (a pod without nodeSelector / node affinity)
        if (T.f_taint[g] && any_taint) return 43;
        bool open = !T.f_affinity[g] || T.f_first[g] == T.f_first[g + 1];
        for (uint32_t i = T.f_first[g]; i < T.f_first[g + 1]; ++i) if (T.f_reqs.defined[i] == 0) open = true;
        if (!open) return 43;
      }
      if (T.min_domains[g] >= 0) return 44;
      const int key = T.key[g], type = T.type[g];
      if (key < 0) {
        if (type == 1 || (inv && type != 2)) return 45;                           // affinity on the hostname
        if (n_host >= kTopoMaxHost) return 46;
        if (type == 0 && (T.max_skew[g] < 1 || T.max_skew[g] > 6)) return 47;     // counters saturate at 7
        if (type == 2 && n_track < kTopoTrack) {
          W::store(&S_.track_field[n_track], (uint32_t)n_host);
          track_fields |= 1ull << (4 * n_host);
          n_track++;
        }
        W::store(&S_.gmap[g], (int16_t)n_host);
        n_host++;
      } else {
        if (inv || type == 2) return 48;                                          // anti-affinity on a dictionary key blocks every value (topology.go:203-206)
        if (n_zg >= kTopoMaxZg) return 49;
        int var = -1;
        for (int j = 0; j < nvv; ++j) if (Mp->vkey[j] == key) var = j;
        if (var < 0 || Mp->vwidth[var] > kTopoMaxDom) return 50;
        const uint32_t w0 = d.key_word_off[key];
        KS_LDS TopoZg& Z = S_.zg[n_zg];
        const int32_t* c0 = T.counts0 + (size_t)g * T.dom_words * 64;
        const uint16_t* rk = T.value_rank + (size_t)w0 * 64;
        W::for_n(kTopoMaxDom, [&](int z) { Z.cnt[z] = c0[z]; Z.rank[z] = rk[z]; });
        if (W::leader()) {
          Z.dom = (uint32_t)T.domains0[(size_t)g * T.dom_words]; Z.nonzero = T.nonzero0[g]; Z.skew = T.max_skew[g];
          Z.type = (uint8_t)type; Z.var = (uint8_t)var;
        }
        // (domains beyond the sixteen the field holds cannot exist: the field's width is the key's highest valid value)
        if (T.domains0[(size_t)g * T.dom_words] >> kTopoMaxDom) return 50;
        W::store(&S_.gmap[g], (int16_t)(0x100 + n_zg));
        n_zg++;
      }
    }
    W::sync();
    // classes
    const int nc = P.n_classes, words = T.words;
    TopoClass* tc = Tk->cls;
    const uint64_t bad = W::reduce_or(nc, [&](int c) -> uint64_t {
      const uint64_t* ct = T.cls_topo + (size_t)c * 2 * words;
      TopoClass k;
      k.hlim = kTopoGuard | (kTopoOnes * 7); k.hinc = 0; k.zsel = 0; k.zg = -1; k.zself = 0; k.excl = 0xFFu; k.pad = 0;
      uint64_t b = 0;
      for (int w = 0; w < words; ++w) {
        const uint64_t ow = ct[w], se = ct[words + w], inv = T.inverse_mask[w];
        // getMatchingTopologies (topology.go:561-574): the groups the pod owns + the inverse groups that select it
        for (uint64_t m = (ow & ~inv) | (se & inv); m; m &= m - 1) {
          const int g = w * 64 + ctz64(m);
          const int gm = S_.gmap[g];
          const bool self = (se >> (g & 63)) & 1;
          if (gm < 0x100) {
            const int lim = T.type[g] == 2 ? 0 : T.max_skew[g] - (self ? 1 : 0);
            if (lim < 0 || lim > 6) { b = 1; continue; }
            const int sh = 4 * gm;
            const uint64_t cur = (k.hlim >> sh) & 7;
            if ((uint64_t)lim < cur) k.hlim = (k.hlim & ~(7ull << sh)) | ((uint64_t)lim << sh);
            if (lim == 0 && T.type[g] == 2 && k.excl == 0xFFu)
              for (int t = 0; t < kTopoTrack; ++t) if (S_.track_field[t] == (uint32_t)gm) k.excl = (uint32_t)t;
          } else {
            if (k.zg >= 0) { b = 1; continue; }   // two groups on dictionary keys: each narrows from the claim's own set (topology.go:226-250)
            k.zg = gm - 0x100; k.zself = self ? 1u : 0u;
          }
        }
        // Record (topology.go:197-220): the regular groups that select the pod, the inverse groups it owns
        for (uint64_t m = (se & ~inv) | (ow & inv); m; m &= m - 1) {
          const int g = w * 64 + ctz64(m);
          const int gm = S_.gmap[g];
          if (gm < 0x100) k.hinc |= 1ull << (4 * gm);
          else k.zsel |= 1ull << (gm - 0x100);
        }
      }
      tc[c] = k;
      return b;
    });
    if (bad) return 51;
    return 0;
  }

  // ---- the pod's domain choice on the dictionary-key group it owns, before any claim is looked at ----
  // vm: candidate domains; st->zkey[z]: priority of domain z (smaller wins; spread: count, then name rank — topologygroup.go:251-297
  // with the canonical tie-break of DESIGN.md §2); multi: every candidate the claim admits stays (affinity with pods to be affine to,
  // topologygroup.go:345-364). false: no domain can satisfy the pod wherever it goes.
  struct ZChoice { uint32_t vm; int off, width; bool multi, on; uint64_t clear; };
  KS_DEV ZChoice choose_domains(const TopoClass& tc, const FastSlot& cs, bool* possible) {
    ZChoice zc; zc.vm = 0; zc.off = 0; zc.width = 0; zc.multi = false; zc.on = false; zc.clear = 0;
    *possible = true;
    if (tc.zg < 0) return zc;
    KS_LDS TopoZg& Z = st->zg[tc.zg];
    const int j = Z.var;
    const int off = Mp->voff[j], width = Mp->vwidth[j];
    const uint32_t fmn = (1u << width) - 1;
    const uint32_t podf = (uint32_t)(cs.cvmask >> off) & fmn;   // the values the pod itself admits (every one when it does not select on the key)
    const uint32_t D = Z.dom;
    const bool self = tc.zself != 0;
    zc.on = true; zc.off = off; zc.width = width;
    zc.clear = Mp->fmask[j] | (1ull << (off + width));
    KS_LDS uint64_t* zk = st->zkey;
    KS_LDS TopoZg* Zp = &Z;
    if (Z.type == 0) {
      // domainMinCount over the domains the pod supports (topologygroup.go:300-322), then count + self - min <= maxSkew
      const uint32_t sup = D & podf;
      const uint64_t mn64 = W::reduce_min(kTopoMaxDom, [&](int z) -> uint64_t { return ((sup >> z) & 1) ? (uint64_t)(uint32_t)Zp->cnt[z] : ~0ull; });
      const long long mn = mn64 == ~0ull ? (long long)INT32_MAX : (long long)mn64;
      const long long skew = Z.skew;
      const uint64_t vb = W::ballot([&](int z) {
        if (z >= kTopoMaxDom) return false;
        const bool v = ((D >> z) & 1) && (long long)Zp->cnt[z] + (self ? 1 : 0) - mn <= skew;
        zk[z] = v ? (((uint64_t)(uint32_t)(Zp->cnt[z] + (self ? 1 : 0)) << 32) | ((uint64_t)Zp->rank[z] << 8) | (uint64_t)z) : ~0ull;
        return v;
      });
      zc.vm = (uint32_t)vb;
      W::sync();
      if (!zc.vm) *possible = false;
      return zc;
    }
    // affinity (topologygroup.go:324-388)
    const uint32_t pn = (uint32_t)W::ballot([&](int z) { return z < kTopoMaxDom && ((D >> z) & 1) && Zp->cnt[z] > 0 && ((podf >> z) & 1); });
    if (pn) { zc.vm = pn; zc.multi = true; return zc; }
    if (!self) { *possible = false; return zc; }
    // nothing to be affine to yet and the pod matches its own selector: the first domain the claim and the pod admit (:372-386; the
    // second loop's pick is the same domain whenever it lies inside the claim's set, and drops out of the intersection otherwise)
    const uint32_t ph = D & podf;
    W::each([&](int z) { if (z < kTopoMaxDom) zk[z] = ((ph >> z) & 1) ? (((uint64_t)Zp->rank[z] << 8) | (uint64_t)z) : ~0ull; });
    W::sync();
    zc.vm = ph;
    if (!ph) *possible = false;
    return zc;
  }

  // CanAdd (nodeclaim.go:124-242) of the class on the claim whose record this lane holds -> bit 0: accepts by the first probe of the
  // requirement-set cache, bit 1: the set is not at its first probe (or has further Pareto vectors): undecided. m2 = the narrowed set.
  KS_DEV int lane_test(const TopoRec& r, const TopoClass& tc, const FastSlot& cs, const ZChoice& zc, uint64_t& m2) const {
    const int t = (int)(r.vmask >> 56);
    const uint64_t m = r.vmask & cs.cvmask;
    m2 = m;
    bool ok = ((cs.tmplok >> t) & 1u) != 0 && fast_fields_ok(m, cs.dmask);
    ok = ok && (((tc.hlim - r.hcnt) & kTopoGuard) == kTopoGuard);
    if (zc.on) {
      const uint32_t zf = (uint32_t)(m >> zc.off) & ((1u << zc.width) - 1);
      const uint32_t cand = zf & zc.vm;
      ok = ok && cand != 0;
      uint32_t nf = cand;
      if (!zc.multi) {
        uint64_t best = ~0ull;
        for (int z = 0; z < zc.width; ++z) { const uint64_t kz = st->zkey[z]; if (((cand >> z) & 1) && kz < best) best = kz; }
        nf = 1u << (best & 0xFF & (kTopoMaxDom - 1));
      }
      m2 = (m & ~zc.clear) | ((uint64_t)nf << zc.off);
    }
    if (!ok) return 0;
    const FastEnt e = lds_get16(&cold.ent[fast_hash(m2)]);
    if (e.vmask != m2) return 2;
    return fast_fits_first(e, r.req, cs.size) ? 1 : 0;
  }
  // the lanes of `todo` the long way: every probe of the cache, every Pareto vector; a set that is not cached gets its entry.
  // Returns the lanes that accept; bail != 0: stop.
  KS_COLD uint64_t resolve(uint64_t todo, LaneVar<uint64_t>& m2v, LaneVar<int32_t>& q0, LaneVar<int32_t>& q1, LaneVar<int32_t>& q2, LaneVar<int32_t>& q3, const FastSlot& cs) {
    uint64_t acc = 0;
    while (todo) {
      const uint64_t td = todo;
      uint64_t okb = 0, miss = 0;
      W::ballot2([&](int l) {
        if (!((td >> l) & 1)) return 0;
        FastEnt e;
        if (fast_lookup(cold.ent, m2v.at(l), e) < 0) return 2;
        const int32_t rq[4] = {q0.at(l), q1.at(l), q2.at(l), q3.at(l)};
        return fast_fits(cold.pool, e, rq, cs.size) ? 1 : 0;
      }, okb, miss);
      acc |= okb;
      todo = miss;
      if (miss && cold.create_entry(m2v.bcast(ctz64(miss))) < 0) { bail = 60; return 0; }
    }
    return acc;
  }

  // Record (topology.go:197-220) on the dictionary-key groups that count the pod: only once the claim is down to ONE domain
  KS_DEV void record_zonal(const TopoClass& tc, uint64_t m2) {
    for (uint64_t zs = tc.zsel; zs; zs &= zs - 1) {
      KS_LDS TopoZg& Z = st->zg[ctz64(zs)];
      const int j = Z.var, off = Mp->voff[j], width = Mp->vwidth[j];
      if ((m2 >> (off + width)) & 1) continue;   // the claim does not define the key: Exists, no values
      const uint32_t f = (uint32_t)(m2 >> off) & ((1u << width) - 1);
      if (popc64(f) != 1) continue;
      const int z = ctz64(f);
      if (W::leader()) { const int32_t c = Z.cnt[z]; Z.cnt[z] = c + 1; if (c == 0) Z.nonzero = Z.nonzero + 1; Z.dom = Z.dom | f; }
    }
    W::sync();
  }
  KS_FN static uint64_t host_add(uint64_t hcnt, uint64_t hinc) {   // per-field +1, saturating at 7
    const uint64_t full = hcnt & (hcnt >> 1) & (hcnt >> 2) & kTopoOnes;
    return hcnt + (hinc & ~full);
  }
  // a claim that now holds a member of the anti-affinity groups in `fields` leaves their lists
  KS_COLD void lists_remove(uint64_t fields, uint32_t x) {
    KS_LDS TopoState& S_ = *st;
    for (int t = 0; t < kTopoTrack; ++t) {
      const uint32_t f = S_.track_field[t];
      if (f == 0xFFu || !((fields >> (4 * f)) & 1)) continue;
      const int n = S_.n_free[t];
      KS_LDS uint32_t* fl = S_.freel[t];
      const int i = W::find_first(0, n, [&](int q) { return fl[q] == x; });
      if (i < n) { if (W::leader()) { fl[i] = fl[n - 1]; S_.n_free[t] = n - 1; } W::sync(); }
    }
  }
  // ... and a new claim joins the lists of the groups it holds no member of
  KS_COLD void lists_add(uint64_t hcnt, uint32_t x) {
    KS_LDS TopoState& S_ = *st;
    for (int t = 0; t < kTopoTrack; ++t) {
      const uint32_t f = S_.track_field[t];
      if (f == 0xFFu || ((hcnt >> (4 * f)) & 7)) continue;
      const int n = S_.n_free[t];
      if (n >= kTopoFreeCap) {
        // too many claims without a member: the list is dropped and the classes it served scan the order like every other class
        if (W::leader()) { S_.track_field[t] = 0xFFu; S_.n_free[t] = 0; }
        track_fields &= ~(1ull << (4 * f));
      } else if (W::leader()) { S_.freel[t][n] = x; S_.n_free[t] = n + 1; }
      W::sync();
    }
  }

  // addToNewNodeClaim (scheduler.go:695-790) for a pod no in-flight claim accepted: 1 = claim created, 0 = stop (bail; -1 = capacity)
  KS_COLD int new_claim(const TopoClass& tc, const FastSlot& cs, const ZChoice& zc, int bi) {
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const int T = P.n_templates, nr = P.n_res, iw = P.it_words;
    const int n = order.n;
    n_ref += (unsigned long long)n;
    for (int t = 0; t < T; ++t) {
      if (!((cold.active_templates >> t) & 1u)) continue;
      const uint32_t lm = P.tmpl_limit_mask[t];
      if (lm) {
        // filterByRemainingResources (scheduler.go:1069-1085): this engine only continues while no type is excluded
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        if (((lm >> nr) & 1) && rem[nr] <= 0) { bail = 23; return 0; }
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
        if (excluded) { bail = 24; return 0; }
      }
      cold.host_seq++;
      n_ref++;
      // CanAdd on the fresh claim: the template's set, no requests, every hostname counter zero (the limits are >= 0)
      TopoRec fresh; fresh.vmask = Mp->tvmask[t]; fresh.req[0] = fresh.req[1] = fresh.req[2] = fresh.req[3] = 0; fresh.hcnt = 0;
      uint64_t m2 = 0;
      // (every lane computes the same verdict: the record is wave-uniform)
      int v = lane_test(fresh, tc, cs, zc, m2);
      v = fast_uniform(v);
      m2 = W::uniform(m2);
      if (v == 0) continue;
      FastEnt e;
      int eh = fast_lookup(cold.ent, m2, e);
      if (eh < 0) { eh = cold.create_entry(m2); if (eh < 0) { bail = 25; return 0; } e = lds_get(&cold.ent[eh]); }
      if (!fast_fits(cold.pool, e, fresh.req, cs.size)) continue;
      if (cold.n_claims >= S.max_claims) { bail = -1; return 0; }
      const int c = cold.n_claims++;
      TopoRec nrq;
      nrq.vmask = m2;
      for (int q = 0; q < 4; ++q) nrq.req[q] = cs.size[q];
      nrq.hcnt = host_add(0, tc.hinc);
      if (W::leader()) { Tk->rec[c] = nrq; F.c_hostseq[c] = cold.host_seq; }
      order.append(c);
      record_zonal(tc, m2);
      if (track_fields) lists_add(nrq.hcnt, (uint32_t)c);
      if (W::leader()) { st->blk_claim[bi] = (uint32_t)c; st->blk_cnt[bi] = 0; }
      W::sync();
      if (lm) {
        // subtractMax (scheduler.go:1049-1066) over the claim's instance types: F(m2) ∩ fits(size)
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
    bail = 27;   // an unschedulable pod: error codes, diagnostics and the relaxation ladder are the general engine's
    return 0;
  }

  // the claims at positions p0 .. p0+63 of the order, one per lane (0xFFFFFFFF past the end); (k, i) = run and index inside it of
  // position p0, moved on to position p0 + 64
  KS_DEV void window(int& k, uint32_t& i, LaneVar<uint32_t>& xv) {
    int filled = 0;
    W::each([&](int l) { xv.at(l) = 0xFFFFFFFFu; });
    const uint32_t* ring = order.ring;
    while (filled < 64 && k <= order.max_cnt) {
      const uint32_t sz = (uint32_t)fast_uniform((int)order.size_(k));
      if (i < sz) {
        const uint32_t left = sz - i;
        const int take = left < (uint32_t)(64 - filled) ? (int)left : 64 - filled;
        const uint32_t h = (uint32_t)fast_uniform((int)order.head_(k)), m = (uint32_t)fast_uniform((int)order.mask_of(k)), o = (uint32_t)fast_uniform((int)order.off_(k)), i0 = i;
        const int f0 = filled;
        W::each([&](int l) { if (l >= f0 && l < f0 + take) xv.at(l) = ring[o + ((h + i0 + (uint32_t)(l - f0)) & m)]; });
        filled += take;
        i += (uint32_t)take;
        if (i < sz) break;
      }
      k++; i = 0;
    }
  }

  KS_COLD void finish(int status, unsigned long long steps) {
    const Workspace& S = *Sk; const FastWork& F = *Fk;
    if (status != 3 && status != 1) {
      const int n = cold.n_claims;
      const int dc = order.defect_claim;
      const bool dapp = order.defect_append;
      order.write_final();   // the array form: o_key (pod counts), o_ord (claims) by position; the last move stays undone, as in the reference
      FastClaim* gs = F.c_state; uint32_t* gn = F.c_npods; uint16_t* ge = F.c_ent;
      const TopoRec* rec = Tk->rec;
      const uint32_t* cnt = order.cnt;
      const KS_LDS FastEnt* en = cold.ent;
      W::for_n(n, [&](int c) {
        const TopoRec r = rec[c];
        FastClaim fc; fc.vmask = r.vmask; for (int q = 0; q < 4; ++q) fc.req[q] = r.req[q];
        FastEnt e;
        gs[c] = fc;
        ge[c] = (uint16_t)fast_lookup(en, r.vmask, e);
        gn[c] = cnt[c] + ((c == dc && !dapp) ? 1u : 0u);
      });
      W::store(S.n_claims_out, n);
    }
    if (status) W::store(S.status_out, status);
    Counters c{};
    c.bin_evaluations = n_tests; c.full_evaluations = n_windows; c.queue_pops = steps; c.sorts = steps; c.slow_sorts = order.slow_sorts;
    c.ref_bin_evaluations = n_ref; c.it_evaluations = n_listed;
    c.cycles[20] = (unsigned long long)(bail > 0 ? bail : 0);
    if (W::leader()) *S.counters = c;
    W::sync();
  }

  // the block's class records, wave-uniform (LDS reads leave the compiler believing they are per-lane values)
  KS_DEV TopoClass class_of(int bi) const {
    TopoClass t = lds_get(&st->blk_tc[bi]);
    t.hlim = W::uniform(t.hlim); t.hinc = W::uniform(t.hinc); t.zsel = W::uniform(t.zsel);
    t.zg = fast_uniform(t.zg); t.zself = (uint32_t)fast_uniform((int)t.zself); t.excl = (uint32_t)fast_uniform((int)t.excl); t.pad = 0;
    return t;
  }
  KS_DEV FastSlot slot_of(int bi) const {
    FastSlot c = lds_get(&st->blk_cs[bi]);
    c.cvmask = W::uniform(c.cvmask); c.dmask = W::uniform(c.dmask);
    for (int q = 0; q < 4; ++q) c.size[q] = fast_uniform(c.size[q]);
    c.tmplok = (uint32_t)fast_uniform((int)c.tmplok); c.kdef = (uint32_t)fast_uniform((int)c.kdef);
    return c;
  }

  KS_DEV void solve() {
    {
      const int why = (int)W::uniform((uint64_t)(uint32_t)cold.setup(true));
      if (why) { bail = why; finish(3, 0); return; }
      const int why2 = (int)W::uniform((uint64_t)(uint32_t)setup_topo());
      if (why2) { bail = why2; finish(3, 0); return; }
    }
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const int np = P.n_pods;
    const uint32_t* gqcls = F.q_class; uint32_t* gqclaim = F.q_claim; uint32_t* gqcnt = F.q_cnt;
    const TopoClass* gtc = Tk->cls; const FastSlot* gcs = F.cls;
    TopoRec* rec = Tk->rec;
    const uint32_t* ocnt = order.cnt;
    KS_LDS TopoState& S_ = *st;
    const volatile int* cancel = S.cancel_flag;
    const long long max_steps = S.max_steps;
    unsigned long long steps = 0;
    int status = 0;
    for (int base = 0; base < np && !status; base += 64) {
      const int bn = np - base < 64 ? np - base : 64;
      // the block's classes with their records: one gather per 64 pods
      W::each([&](int l) {
        if (l < bn) {
          const uint32_t k = gqcls[base + l] & ~kFastLastBit;
          S_.blk_cls[l] = k; lds_put(&S_.blk_tc[l], gtc[k]); lds_put(&S_.blk_cs[l], gcs[k]);
          S_.blk_claim[l] = 0xFFFFFFFFu; S_.blk_cnt[l] = 0;
        }
      });
      W::sync();
      if (cancel) {
        // > 0: ksolve_cancel / the deadline; < 0 (tests only, KSOLVE_TEST_CANCEL_AT): as if the cancel landed once -flag pods were placed
        const int cv = fast_uniform((int)W::poll_flag(cancel));
        if (cv > 0 || (cv < 0 && (long long)steps >= -(long long)cv)) { status = 2; break; }
      }
      int bi = 0;
      for (; bi < bn; ++bi) {
        if (max_steps >= 0 && (long long)steps >= max_steps) { status = 2; break; }
        steps++;
        order.sort();                                      // scheduler.go:598: the move the last commit left behind
        if (order.overflow) { status = 1; break; }
        const TopoClass tc = class_of(bi);
        const FastSlot cs = slot_of(bi);
        bool possible = true;
        const ZChoice zc = choose_domains(tc, cs, &possible);
        if (!possible) { bail = 27; status = 3; break; }
        const int n = order.n;
        // ---- addToInflightNode (scheduler.go:658-692): the first claim of the order that accepts ----
        LaneVar<uint64_t> m2v, hcv;
        LaneVar<uint32_t> xv, pv, cv;
        LaneVar<int32_t> q0, q1, q2, q3;
        uint64_t okm = 0;
        bool found = false;
        // the acceptor (wave-uniform): claim, position, its narrowed requirement set, hostname counters, requests, pod count
        uint32_t kx = 0, a_pos = 0, kc = 0; uint64_t km = 0, kh = 0; int32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        const bool listed = tc.excl != 0xFFu && fast_uniform((int)S_.track_field[tc.excl & (kTopoTrack - 1)]) != 0xFF;
        if (listed) {
          // every member of an anti-affinity group it owns needs a claim without one: the claims that hold none yet, wherever they
          // stand in the order — the lowest position among those that accept
          const int t = (int)tc.excl;
          const int nf = fast_uniform((int)S_.n_free[t]);
          n_listed++;
          uint32_t best_pos = 0xFFFFFFFFu;
          for (int f0 = 0; f0 < nf; f0 += 64) {
            uint64_t und = 0;
            W::ballot2([&](int l) {
              xv.at(l) = 0xFFFFFFFFu; pv.at(l) = 0xFFFFFFFFu;
              if (f0 + l >= nf) return 0;
              const uint32_t x = S_.freel[t][f0 + l];
              const TopoRec r = rec[x];
              uint64_t m2;
              const int v = lane_test(r, tc, cs, zc, m2);
              xv.at(l) = x; m2v.at(l) = m2; hcv.at(l) = r.hcnt; pv.at(l) = order.position((int)x); cv.at(l) = ocnt[x];
              q0.at(l) = r.req[0]; q1.at(l) = r.req[1]; q2.at(l) = r.req[2]; q3.at(l) = r.req[3];
              return v;
            }, okm, und);
            n_tests += (unsigned long long)(nf - f0 < 64 ? nf - f0 : 64);
            if (und) { okm |= resolve(und, m2v, q0, q1, q2, q3, cs); if (bail) break; }
            if (okm) {
              int who = -1;
              const uint64_t om = okm;
              const uint32_t p = W::argmin_u32([&](int l) { return ((om >> l) & 1) ? pv.at(l) : 0xFFFFFFFFu; }, &who);
              if (p < best_pos) {
                best_pos = p; found = true; a_pos = p;
                kx = xv.bcast(who); km = m2v.bcast(who); kh = hcv.bcast(who); kc = cv.bcast(who);
                k0 = q0.bcast(who); k1 = q1.bcast(who); k2 = q2.bcast(who); k3 = q3.bcast(who);
              }
            }
          }
          if (bail) { status = 3; break; }
        } else {
          int k = 1; uint32_t i = 0;
          for (int p0 = 0; p0 < n; p0 += 64) {
            window(k, i, xv);
            n_windows++;
            uint64_t und = 0;
            W::ballot2([&](int l) {
              const uint32_t x = xv.at(l);
              if (x == 0xFFFFFFFFu) return 0;
              const TopoRec r = rec[x];
              uint64_t m2;
              const int v = lane_test(r, tc, cs, zc, m2);
              m2v.at(l) = m2; hcv.at(l) = r.hcnt; cv.at(l) = ocnt[x];
              q0.at(l) = r.req[0]; q1.at(l) = r.req[1]; q2.at(l) = r.req[2]; q3.at(l) = r.req[3];
              return v;
            }, okm, und);
            n_tests += (unsigned long long)(n - p0 < 64 ? n - p0 : 64);
            // the lanes in front of the first plain acceptor whose requirement set is not at its first probe: the long way
            const uint64_t before = okm ? (und & ((1ull << ctz64(okm)) - 1)) : und;
            if (before) { okm |= resolve(before, m2v, q0, q1, q2, q3, cs); if (bail) break; }
            if (okm) {
              const int a = ctz64(okm);
              found = true; a_pos = (uint32_t)(p0 + a);
              kx = xv.bcast(a); km = m2v.bcast(a); kh = hcv.bcast(a); kc = cv.bcast(a);
              k0 = q0.bcast(a); k1 = q1.bcast(a); k2 = q2.bcast(a); k3 = q3.bcast(a);
              break;
            }
          }
          if (bail) { status = 3; break; }
        }
        if (!found) {
          const int made = fast_uniform(new_claim(tc, cs, zc, bi));
          if (!made) { status = bail < 0 ? 1 : 3; break; }
          continue;
        }
        // ---- NodeClaim.Add (nodeclaim.go:247-263) ----
        n_ref += (unsigned long long)a_pos + 1;
        TopoRec nrq;
        nrq.vmask = km;
        nrq.req[0] = k0 + cs.size[0]; nrq.req[1] = k1 + cs.size[1]; nrq.req[2] = k2 + cs.size[2]; nrq.req[3] = k3 + cs.size[3];
        nrq.hcnt = host_add(kh, tc.hinc);
        if (W::leader()) { rec[kx] = nrq; S_.blk_claim[bi] = kx; S_.blk_cnt[bi] = kc; }
        record_zonal(tc, km);
        // the anti-affinity lists: the claim leaves those whose counter this pod takes from zero
        {
          const uint64_t zero_before = ~(kh | (kh >> 1) | (kh >> 2)) & kTopoOnes;
          const uint64_t leaving = tc.hinc & zero_before & track_fields;
          if (leaving) lists_remove(leaving, kx);
        }
        order.defect = (int)a_pos; order.defect_claim = (int)kx; order.defect_append = false;   // RunOrder::increment: it moves at the next sort
        W::sync();
      }
      // the block's results, in queue order (ksolve_fast_scatter puts them under the pod indices)
      {
        const int dn = bi < bn ? bi : bn;
        W::each([&](int l) { if (l < dn) { gqclaim[base + l] = S_.blk_claim[l]; gqcnt[base + l] = S_.blk_cnt[l]; } });
        W::sync();
      }
    }
    finish(status, steps);
  }
};

}  // namespace ks
