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
// Widened after its first GPU passes (round 6): minDomains (topologygroup.go:318-320), up to TWO groups on dictionary keys per pod
// (zone + capacity-type spread; spread and affinity on one key — each narrows from the claim's own set, topology.go:226-250; such a
// pod takes the out-of-line select so that the loop's own step carries neither the code nor the registers), pod anti-affinity on a
// dictionary key with its inverse groups (every domain the pod could land in is blocked, topology.go:203-206, :213-219).
//
// Whatever it does not handle (node filters, groups created by relaxation, affinity on the hostname, three or more dictionary-key
// groups on one pod, maxSkew beyond a counter's range, an unschedulable pod, NodePool limits that exclude a type ...) makes it stop
// with status 3 before it has written a result; the host runs the general engine on the same problem. There is no CPU path.
#pragma once
#include "fast_engine.h"
#include "run_order.h"
#include "topo_types.h"

namespace ks {

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
  unsigned long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tc4 = 0, tc5 = 0, tc6 = 0, tc7 = 0, tc8 = 0, tc9 = 0;   // profiling builds (-DKSOLVE_PHASE_TIMERS): shader clock per phase of a step
  int last_kind = 0, last_x = 0, last_p = 0;   // the move of the last step (1: a claim gained a pod, from position last_p; 2: a new claim), 0: pending in `order` / none

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
        if (inv && type != 2) return 48;
        if (type == 0 && (T.max_skew[g] < 1 || T.max_skew[g] > 30000 || T.min_domains[g] > 30000)) return 44;
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
          Z.dom = (uint32_t)T.domains0[(size_t)g * T.dom_words]; Z.nonzero = T.nonzero0[g]; Z.skew = (int16_t)(type == 0 ? T.max_skew[g] : 0); Z.min_domains = (int16_t)(type == 0 ? T.min_domains[g] : -1);
          Z.type = (uint8_t)type; Z.var = (uint8_t)var; Z.off = Mp->voff[var]; Z.width = Mp->vwidth[var];
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
      k.hlim = kTopoGuard | (kTopoOnes * 7); k.hinc = 0; k.zsel = 0; k.zg[0] = -1; k.zg[1] = -1; k.zself = 0; k.excl = 0xFFu; k.pad = 0;
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
            // (up to two groups on dictionary keys: each narrows from the claim's own set, the intersection stands — topology.go:226-250)
            const int slot = k.zg[0] < 0 ? 0 : k.zg[1] < 0 ? 1 : -1;
            if (slot < 0) { b = 1; continue; }
            k.zg[slot] = (int16_t)(gm - 0x100); if (self) k.zself |= 1u << slot;
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
  // vm: candidate domains; zkv: priority of domain z in lane z (smaller wins; spread: count, then name rank — topologygroup.go:251-297
  // with the canonical tie-break of DESIGN.md §2); multi: every candidate the claim admits stays (affinity with pods to be affine to,
  // topologygroup.go:345-364). possible = false: no domain can satisfy the pod wherever it goes. Registers and LDS only.
  struct ZChoice { uint32_t vm; int off, width; bool multi, on; uint64_t clear; };
  KS_DEV ZChoice choose_domains(int zg, bool self, uint64_t cvmask, LaneVar<uint64_t>& zkv, bool& possible) const {
    ZChoice zc; zc.vm = 0; zc.off = 0; zc.width = 0; zc.multi = false; zc.on = false; zc.clear = 0;
    possible = true;
    if (zg < 0) return zc;
    KS_LDS TopoZg* const Z = &st->zg[zg];
    TopoZgHead hd = lds_get16((const KS_LDS TopoZgHead*)&Z->dom);
    const uint32_t D = (uint32_t)fast_uniform((int)hd.dom);
    const int skew = fast_uniform((int)hd.skew), mind = fast_uniform((int)hd.min_domains);
    const uint32_t tv = (uint32_t)fast_uniform((int)((uint32_t)hd.type | ((uint32_t)hd.off << 8) | ((uint32_t)hd.width << 16)));
    const int type = (int)(tv & 0xFF), off = (int)((tv >> 8) & 0xFF), width = (int)((tv >> 16) & 0xFF);
    const uint32_t fmn = (1u << width) - 1;
    const uint32_t podf = (uint32_t)(cvmask >> off) & fmn;   // the values the pod itself admits (every one when it does not select on the key)
    zc.on = true; zc.off = off; zc.width = width;
    zc.clear = ((uint64_t)fmn << off) | (1ull << (off + width));
    LaneVar<int32_t> cz; LaneVar<uint32_t> rz;
    W::each([&](int z) { cz.at(z) = Z->cnt[z & (kTopoMaxDom - 1)]; rz.at(z) = Z->rank[z & (kTopoMaxDom - 1)]; });
    if (type == 0) {
      // domainMinCount over the domains the pod supports (topologygroup.go:300-322), then count + self - min <= maxSkew
      const uint32_t sup = D & podf;
      int who = 0;
      const uint32_t mn32 = W::argmin_u32([&](int z) { return (z < kTopoMaxDom && ((sup >> z) & 1)) ? (uint32_t)cz.at(z) : 0xFFFFFFFFu; }, &who);
      long long mn = mn32 == 0xFFFFFFFFu ? (long long)INT32_MAX : (long long)mn32;
      if (mind >= 0 && popc64(sup) < mind) mn = 0;                               // minDomains — topologygroup.go:318-320
      const uint64_t vb = W::ballot([&](int z) {
        const bool v = z < kTopoMaxDom && ((D >> z) & 1) && (long long)cz.at(z) + (self ? 1 : 0) - mn <= (long long)skew;
        zkv.at(z) = v ? (((uint64_t)(uint32_t)(cz.at(z) + (self ? 1 : 0)) << 32) | ((uint64_t)rz.at(z) << 8) | (uint64_t)z) : ~0ull;
        return v;
      });
      zc.vm = (uint32_t)vb;
      if (!zc.vm) possible = false;
      return zc;
    }
    if (type == 2) {
      // anti-affinity: the empty domains the pod admits (topologygroup.go:404-439); every one the claim admits stays
      zc.vm = (uint32_t)W::ballot([&](int z) { return z < kTopoMaxDom && ((D >> z) & 1) && cz.at(z) == 0 && ((podf >> z) & 1); });
      zc.multi = true;
      if (!zc.vm) possible = false;
      return zc;
    }
    // affinity (topologygroup.go:324-388)
    const uint32_t pn = (uint32_t)W::ballot([&](int z) { return z < kTopoMaxDom && ((D >> z) & 1) && cz.at(z) > 0 && ((podf >> z) & 1); });
    if (pn) { zc.vm = pn; zc.multi = true; return zc; }
    if (!self) { possible = false; return zc; }
    // nothing to be affine to yet and the pod matches its own selector: the first domain the claim and the pod admit (:372-386; the
    // second loop's pick is the same domain whenever it lies inside the claim's set, and drops out of the intersection otherwise)
    const uint32_t ph = D & podf;
    W::each([&](int z) { zkv.at(z) = (z < kTopoMaxDom && ((ph >> z) & 1)) ? (((uint64_t)rz.at(z) << 8) | (uint64_t)z) : ~0ull; });
    zc.vm = ph;
    if (!ph) possible = false;
    return zc;
  }

  // CanAdd (nodeclaim.go:124-242) of a class on the claim whose state this lane holds -> 1: accepts, by the first probe of the
  // requirement-set cache; 2: the set is not at its first probe (or has further Pareto vectors): undecided; 0: rejects. m2 = the set
  // narrowed by the pod's selectors and its domain choice.
  struct KClass { uint64_t hlim, cvmask, dmask; int32_t s0, s1, s2, s3; uint32_t tmplok; };
  // one group's choice on the claim's set: the claim's field of the key ∧ the candidates (∧ what an earlier group of the same key left)
  // FIRST: m2 is still the claim's own set (the first group's choice): the candidates are a subset of its field, nothing to intersect
  template <bool FIRST>
  KS_DEV bool apply_choice(uint64_t m, uint64_t& m2, const ZChoice& zc, const LaneVar<uint64_t>& zkv) const {
    const uint32_t fm = (1u << zc.width) - 1;
    const uint32_t zf = (uint32_t)(m >> zc.off) & fm;
    const uint32_t cand = zf & zc.vm;
    uint32_t nf = cand;
    if (!zc.multi) {
      uint64_t best = ~0ull;
      for (int z = 0; z < zc.width; ++z) { const uint64_t kz = zkv.bcast(z); if (((cand >> z) & 1) && kz < best) best = kz; }
      nf = 1u << (best & (kTopoMaxDom - 1));   // (no candidate: whatever — the verdict below says so)
    }
    if constexpr (FIRST) {
      m2 = (m & ~zc.clear) | ((uint64_t)nf << zc.off);
      return cand != 0;
    } else {
      nf = cand ? nf & ((uint32_t)(m2 >> zc.off) & fm) : 0u;
      m2 = (m2 & ~zc.clear) | ((uint64_t)nf << zc.off);
      return nf != 0;
    }
  }
  // TWO = false: the loop's own form — a class with a second group on a dictionary key goes through the out-of-line paths, so that the
  // step of every other class carries neither its code nor its registers
  template <bool TWO = true>
  KS_DEV int lane_test(uint64_t vmask, int32_t r0, int32_t r1, int32_t r2, int32_t r3, uint64_t hcnt, const KClass& k, const ZChoice& zc, const LaneVar<uint64_t>& zkv,
                       const ZChoice& zd, const LaneVar<uint64_t>& zkw, const KS_LDS FastEnt* ent, uint64_t& m2) const {
    const int t = (int)(vmask >> 56) & 31;   // (template ids are < 32; a lane past the window's end holds anything)
    const uint64_t m = vmask & k.cvmask;
    m2 = m;
    bool ok = ((k.tmplok >> t) & 1u) != 0 && fast_fields_ok(m, k.dmask);
    ok = ok && (((k.hlim - hcnt) & kTopoGuard) == kTopoGuard);
    if (zc.on) ok = apply_choice<true>(m, m2, zc, zkv) && ok;
    if constexpr (TWO) { if (zd.on) ok = apply_choice<false>(m, m2, zd, zkw) && ok; }
    const FastEnt e = lds_get16(&ent[fast_hash(m2)]);   // (read whatever `ok` says: no branch in front of the LDS access)
    if (!ok) return 0;
    if (e.vmask != m2) return 2;
    return (int)((k.s0 <= e.cap[0] - r0) & (k.s1 <= e.cap[1] - r1) & (k.s2 <= e.cap[2] - r2) & (k.s3 <= e.cap[3] - r3));
  }
  // the lanes of `todo` the long way: every probe of the cache, every Pareto vector; a set that is not cached gets its entry.
  // Returns the lanes that accept; bail != 0: stop. (Everything by value: a reference into the caller's registers would pin them to scratch.)
  KS_COLD uint64_t resolve(uint64_t todo, LaneVar<uint64_t> m2v, LaneVar<int32_t> q0, LaneVar<int32_t> q1, LaneVar<int32_t> q2, LaneVar<int32_t> q3, int32_t s0, int32_t s1, int32_t s2, int32_t s3) {
    uint64_t acc = 0;
    const int32_t sz[4] = {s0, s1, s2, s3};
    while (todo) {
      const uint64_t td = todo;
      uint64_t okb = 0, miss = 0;
      W::ballot2([&](int l) {
        if (!((td >> l) & 1)) return 0;
        FastEnt e;
        if (fast_lookup(cold.ent, m2v.at(l), e) < 0) return 2;
        const int32_t rq[4] = {q0.at(l), q1.at(l), q2.at(l), q3.at(l)};
        return fast_fits(cold.pool, e, rq, sz) ? 1 : 0;
      }, okb, miss);
      acc |= okb;
      todo = miss;
      if (miss && cold.create_entry(m2v.bcast(ctz64(miss))) < 0) { bail = 60; return 0; }
    }
    return acc;
  }

  // Record (topology.go:197-220) on the dictionary-key groups that count the pod: only once the claim is down to ONE domain
  KS_DEV void record_zonal(uint64_t zsel, uint64_t m2) const {
    for (uint64_t zs = zsel; zs; zs &= zs - 1) {
      KS_LDS TopoZg* const Z = &st->zg[ctz64(zs)];
      const uint32_t ow = (uint32_t)fast_uniform((int)(*(const KS_LDS uint32_t*)&Z->type));   // type | var << 8 | off << 16 | width << 24
      const int off = (int)((ow >> 16) & 0xFF), width = (int)(ow >> 24);
      if ((m2 >> (off + width)) & 1) continue;   // the claim does not define the key: Exists, no values
      const uint32_t f = (uint32_t)(m2 >> off) & ((1u << width) - 1);
      if ((ow & 0xFF) == 2) {
        // anti-affinity (and its inverse groups) blocks every domain the pod could land in (topology.go:203-206, :213-219)
        if (W::leader()) { int fresh = 0; for (uint32_t b = f; b; b &= b - 1) { const int z = ctz64(b); const int32_t c = Z->cnt[z]; Z->cnt[z] = c + 1; fresh += c == 0; } Z->nonzero = Z->nonzero + fresh; Z->dom = Z->dom | f; }
        continue;
      }
      if (popc64(f) != 1) continue;
      const int z = ctz64(f);
      if (W::leader()) { const int32_t c = Z->cnt[z]; Z->cnt[z] = c + 1; if (c == 0) Z->nonzero = Z->nonzero + 1; Z->dom = Z->dom | f; }
    }
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

  // addToNewNodeClaim (scheduler.go:695-790) for a pod no in-flight claim accepted, every case: several templates, NodePool limits,
  // requirement sets that are not cached yet, pdqsort's other paths. The claim's id, or -1: stop (bail; -1 = capacity).
  // (solve() places the common case itself: one template without limits, the set cached, the single stable move.)
  KS_COLD int new_claim(TopoClass tc, FastSlot cs) {
    const ProblemView& P = *Pk; const Workspace& S = *Sk; const FastWork& F = *Fk;
    const int T = P.n_templates, nr = P.n_res, iw = P.it_words;
    const int n = order.n;
    n_ref += (unsigned long long)n;
    LaneVar<uint64_t> zkv, zkw;
    bool possible = true, possible2 = true;
    const ZChoice zc = choose_domains(tc.zg[0], (tc.zself & 1u) != 0, cs.cvmask, zkv, possible);
    const ZChoice zd = choose_domains(tc.zg[1], (tc.zself & 2u) != 0, cs.cvmask, zkw, possible2);
    if (!possible || !possible2) { bail = 27; return -1; }
    KClass kc; kc.hlim = tc.hlim; kc.cvmask = cs.cvmask; kc.dmask = cs.dmask; kc.s0 = cs.size[0]; kc.s1 = cs.size[1]; kc.s2 = cs.size[2]; kc.s3 = cs.size[3]; kc.tmplok = cs.tmplok;
    for (int t = 0; t < T; ++t) {
      if (!((cold.active_templates >> t) & 1u)) continue;
      const uint32_t lm = P.tmpl_limit_mask[t];
      if (lm) {
        // filterByRemainingResources (scheduler.go:1069-1085): this engine only continues while no type is excluded
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        if (((lm >> nr) & 1) && rem[nr] <= 0) { bail = 23; return -1; }
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
        if (excluded) { bail = 24; return -1; }
      }
      cold.host_seq++;
      n_ref++;
      // CanAdd on the fresh claim: the template's set, no requests, every hostname counter zero (the limits are >= 0)
      uint64_t m2 = 0;
      int v = lane_test(Mp->tvmask[t], 0, 0, 0, 0, 0ull, kc, zc, zkv, zd, zkw, cold.ent, m2);   // (every lane computes the same verdict: the state is wave-uniform)
      v = fast_uniform(v);
      m2 = W::uniform(m2);
      if (v == 0) continue;
      FastEnt e;
      int eh = fast_lookup(cold.ent, m2, e);
      if (eh < 0) { eh = cold.create_entry(m2); if (eh < 0) { bail = 25; return -1; } e = lds_get(&cold.ent[eh]); }
      const int32_t zero[4] = {0, 0, 0, 0};
      if (!fast_fits(cold.pool, e, zero, cs.size)) continue;
      if (cold.n_claims >= S.max_claims) { bail = -1; return -1; }
      const int c = cold.n_claims++;
      TopoRec nrq;
      nrq.vmask = m2;
      for (int q = 0; q < 4; ++q) nrq.req[q] = cs.size[q];
      nrq.hcnt = host_add(0, tc.hinc);
      if (W::leader()) { Tk->rec[c] = nrq; F.c_hostseq[c] = cold.host_seq; }
      order.append(c);
      if (order.single_move(order.defect)) {   // (its place behind the claims with one pod: now, like every single stable move)
        order.move_appended(c);
        order.defect = -1; order.defect_claim = -1; order.defect_append = false;
        last_kind = 2; last_x = c; last_p = 0;
      } else last_kind = 0;
      record_zonal(tc.zsel, m2);
      if (track_fields) lists_add(nrq.hcnt, (uint32_t)c);
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
      return c;
    }
    bail = 27;   // an unschedulable pod: error codes, diagnostics and the relaxation ladder are the general engine's
    return -1;
  }
  // a claim gained a pod and pdqsort's repair is not the single stable move (or its run lies beyond the LDS tables): through RunOrder
  KS_COLD void move_cold(int x, int k, uint32_t a_pos) {
    if (order.single_move((int)a_pos)) {
      const uint32_t idx = a_pos - order.prefix_(k);
      order.move_known(x, k, idx);
      last_kind = 1; last_x = x; last_p = (int)a_pos;
    } else {
      order.defect = (int)a_pos; order.defect_claim = x; order.defect_append = false;   // RunOrder::increment: at the next sort
      last_kind = 0;
      W::sync();
    }
  }
  // ---- the select step's rare forms, out of line (by value in, by value out: nothing of the loop's registers is pinned) ----
  struct Accept { int found; uint32_t x, pos, cnt; uint64_t m2, hcnt; int32_t q0, q1, q2, q3; };
  struct Win { LaneVar<uint32_t> x, c; LaneVar<uint64_t> v, h; LaneVar<int32_t> q0, q1, q2, q3; int n; };
  // every member of an anti-affinity group the class owns needs a claim without one: the claims of list t — those that hold none yet,
  // wherever they stand in the order — and among those that accept the one at the lowest position
  KS_COLD Accept scan_list(int t, KClass kc, ZChoice zc, LaneVar<uint64_t> zkv, ZChoice zd, LaneVar<uint64_t> zkw) {
    Accept A; A.found = 0; A.x = 0; A.pos = 0; A.cnt = 0; A.m2 = 0; A.hcnt = 0; A.q0 = A.q1 = A.q2 = A.q3 = 0;
    KS_LDS TopoState* const S_ = st;
    KS_LDS RunTables* const RT = order.T;
    const KS_GLOBAL TopoRec* const rec = (const KS_GLOBAL TopoRec*)Tk->rec;
    const KS_GLOBAL uint32_t* const ocnt = (const KS_GLOBAL uint32_t*)order.cnt;
    const KS_GLOBAL uint32_t* const oslot = (const KS_GLOBAL uint32_t*)order.slot;
    const KS_LDS FastEnt* const ent = cold.ent;
    const int nf = (int)W::uniform((uint64_t)(uint32_t)S_->n_free[t]);
    uint32_t best_pos = 0xFFFFFFFFu;
    for (int f0 = 0; f0 < nf; f0 += 64) {
      LaneVar<uint64_t> m2v, hcv;
      LaneVar<uint32_t> xv, pv, cv;
      LaneVar<int32_t> q0, q1, q2, q3;
      uint64_t okm = 0, und = 0;
      W::ballot2([&](int l) {
        const uint32_t x = S_->freel[t][f0 + l < nf ? f0 + l : nf - 1];
        const TopoRec r = load_rec(rec + x);
        const uint32_t c = ocnt[x], sl = oslot[x];
        uint64_t m2;
        const int v = lane_test(r.vmask, r.req[0], r.req[1], r.req[2], r.req[3], r.hcnt, kc, zc, zkv, zd, zkw, ent, m2);
        // position = the run's first position + the claim's index inside the run (run_order.h)
        const RunEnt e = lds_get16(&RT->e[c < (uint32_t)kRunMaxCount ? c : 0]);
        const uint32_t mk = (1u << RT->log2cap[c < (uint32_t)kRunMaxCount ? c : 0]) - 1u;
        xv.at(l) = x; m2v.at(l) = m2; hcv.at(l) = r.hcnt; cv.at(l) = c;
        pv.at(l) = (f0 + l < nf && c < (uint32_t)kRunMaxCount) ? e.prefix + ((sl - e.head) & mk) : 0xFFFFFFFFu;
        q0.at(l) = r.req[0]; q1.at(l) = r.req[1]; q2.at(l) = r.req[2]; q3.at(l) = r.req[3];
        return f0 + l < nf ? v : 0;
      }, okm, und);
      if (W::ballot([&](int l) { return f0 + l < nf && cv.at(l) >= (uint32_t)kRunMaxCount; })) { bail = 61; return A; }   // (a claim beyond the LDS ring tables in a list: not this engine's shape)
      if (und) { okm |= resolve(und, m2v, q0, q1, q2, q3, kc.s0, kc.s1, kc.s2, kc.s3); if (bail) return A; }
      if (okm) {
        int who = -1;
        const uint64_t om = okm;
        const uint32_t p = W::argmin_u32([&](int l) { return ((om >> l) & 1) ? pv.at(l) : 0xFFFFFFFFu; }, &who);
        if (p < best_pos) {
          best_pos = p; A.found = 1; A.pos = p;
          A.x = xv.bcast(who); A.m2 = m2v.bcast(who); A.hcnt = hcv.bcast(who); A.cnt = cv.bcast(who);
          A.q0 = q0.bcast(who); A.q1 = q1.bcast(who); A.q2 = q2.bcast(who); A.q3 = q3.bcast(who);
        }
      }
    }
    return A;
  }
  // the claims at positions p0 .. p0+63 of the order, one per lane (0 past the end), `filled` of them; (k, i) = run and index inside
  // it of position p0, moved on to p0 + 64. The table records of sixteen runs come in one LDS round trip (one lane each).
  KS_DEV int order_window(int& k, uint32_t& i, LaneVar<uint32_t>& xv) {
    KS_LDS RunTables* const RT = order.T;
    const KS_GLOBAL uint32_t* const ring = (const KS_GLOBAL uint32_t*)order.ring;
    const int max_cnt = order.max_cnt;
    int filled = 0;
    bool done = false;
    W::each([&](int l) { xv.at(l) = 0; });
    while (!done && filled < 64 && k <= max_cnt) {
      if (k + 16 >= kRunMaxCount) { bail = 62; return filled; }   // (a run beyond the LDS ring tables at the front of the order: not this engine's shape)
      LaneVar<uint32_t> eh, es, eo, em;
      const int kb = k;
      W::each([&](int l) {
        const int kk = kb + (l & 15);
        const RunEnt e = lds_get16(&RT->e[kk]);
        eh.at(l) = e.head; es.at(l) = e.size; eo.at(l) = e.off; em.at(l) = (1u << RT->log2cap[kk]) - 1u;
      });
      for (int j = 0; j < 16 && k <= max_cnt; ++j) {
        const uint32_t sz = es.bcast(j);
        if (i < sz) {
          const uint32_t left = sz - i;
          const int take = left < (uint32_t)(64 - filled) ? (int)left : 64 - filled;
          const uint32_t h = eh.bcast(j), m = em.bcast(j), o = eo.bcast(j), i0 = i;
          const int f0 = filled;
          W::each([&](int l) { if (l >= f0 && l < f0 + take) xv.at(l) = ring[o + ((h + i0 + (uint32_t)(l - f0)) & m)]; });
          filled += take;
          i += (uint32_t)take;
          if (i < sz || filled >= 64) { done = true; break; }
        }
        k++; i = 0;
      }
    }
    return filled;
  }
  // the front of the order with its records, for the loop's window registers
  KS_COLD Win read_window() {
    Win w;
    const KS_GLOBAL TopoRec* const rec = (const KS_GLOBAL TopoRec*)Tk->rec;
    const KS_GLOBAL uint32_t* const ocnt = (const KS_GLOBAL uint32_t*)order.cnt;
    int k = 1; uint32_t i = 0;
    LaneVar<uint32_t> xv;
    const int fl = order_window(k, i, xv);
    W::each([&](int l) {
      const uint32_t x = xv.shuffle(l, l < fl ? l : (fl > 0 ? fl - 1 : 0));   // (lanes past the order's end read the last claim again: no lane is switched off for the loads)
      const TopoRec r = load_rec(rec + x);
      w.x.at(l) = x; w.c.at(l) = ocnt[x]; w.v.at(l) = r.vmask; w.h.at(l) = r.hcnt;
      w.q0.at(l) = r.req[0]; w.q1.at(l) = r.req[1]; w.q2.at(l) = r.req[2]; w.q3.at(l) = r.req[3];
    });
    w.n = fl;
    return w;
  }
  // no acceptor inside the window and the order goes on beyond it: the rings and the records, 64 positions per step, from the front
  KS_COLD Accept scan_order(KClass kc, ZChoice zc, LaneVar<uint64_t> zkv, ZChoice zd, LaneVar<uint64_t> zkw) {
    Accept A; A.found = 0; A.x = 0; A.pos = 0; A.cnt = 0; A.m2 = 0; A.hcnt = 0; A.q0 = A.q1 = A.q2 = A.q3 = 0;
    const KS_GLOBAL TopoRec* const rec = (const KS_GLOBAL TopoRec*)Tk->rec;
    const KS_GLOBAL uint32_t* const ocnt = (const KS_GLOBAL uint32_t*)order.cnt;
    const KS_LDS FastEnt* const ent = cold.ent;
    const int n = order.n;
    int k = 1; uint32_t i = 0;
    for (int p0 = 0; p0 < n; p0 += 64) {
      LaneVar<uint32_t> xv;
      const int fl = order_window(k, i, xv);
      if (bail) return A;
      n_windows++;
      LaneVar<uint64_t> m2v, hcv;
      LaneVar<uint32_t> cv;
      LaneVar<int32_t> q0, q1, q2, q3;
      uint64_t okm = 0, und = 0;
      W::ballot2([&](int l) {
        const uint32_t x = xv.shuffle(l, l < fl ? l : (fl > 0 ? fl - 1 : 0));
        const TopoRec r = load_rec(rec + x);
        const uint32_t c = ocnt[x];
        uint64_t m2;
        const int v = lane_test(r.vmask, r.req[0], r.req[1], r.req[2], r.req[3], r.hcnt, kc, zc, zkv, zd, zkw, ent, m2);
        m2v.at(l) = m2; hcv.at(l) = r.hcnt; cv.at(l) = c;
        q0.at(l) = r.req[0]; q1.at(l) = r.req[1]; q2.at(l) = r.req[2]; q3.at(l) = r.req[3];
        return l < fl ? v : 0;
      }, okm, und);
      // the lanes in front of the first plain acceptor whose requirement set is not at its first probe: the long way
      const uint64_t before = okm ? (und & ((1ull << ctz64(okm)) - 1)) : und;
      if (before) { okm |= resolve(before, m2v, q0, q1, q2, q3, kc.s0, kc.s1, kc.s2, kc.s3); if (bail) return A; }
      if (okm) {
        const int a = ctz64(okm);
        A.found = 1; A.pos = (uint32_t)(p0 + a);
        A.x = xv.bcast(a); A.m2 = m2v.bcast(a); A.hcnt = hcv.bcast(a); A.cnt = cv.bcast(a);
        A.q0 = q0.bcast(a); A.q1 = q1.bcast(a); A.q2 = q2.bcast(a); A.q3 = q3.bcast(a);
        return A;
      }
    }
    return A;
  }
  // the select step of a class with a second group on a dictionary key: its choice, then the list (list_t >= 0) or the whole order
  KS_COLD Accept scan_two(int list_t, KClass kc, ZChoice zc, LaneVar<uint64_t> zkv, int zg1, bool self1) {
    LaneVar<uint64_t> zkw;
    bool p2 = true;
    const ZChoice zd = choose_domains(zg1, self1, kc.cvmask, zkw, p2);
    if (!p2) { bail = 27; Accept A; A.found = 0; A.x = 0; A.pos = 0; A.cnt = 0; A.m2 = 0; A.hcnt = 0; A.q0 = A.q1 = A.q2 = A.q3 = 0; return A; }
    return list_t >= 0 ? scan_list(list_t, kc, zc, zkv, zd, zkw) : scan_order(kc, zc, zkv, zd, zkw);
  }
  KS_COLD void sort_cold() { order.sort(); }

  KS_COLD void finish(int status, unsigned long long steps) {
    const Workspace& S = *Sk; const FastWork& F = *Fk;
    if (status != 3 && status != 1) {
      const int n = cold.n_claims;
      const int dc = order.defect_claim;
      const bool dapp = order.defect_append;
      order.write_final();   // the array form: o_key (pod counts), o_ord (claims) by position; the last move stays undone, as in the reference
      if (last_kind && dc < 0) {
        // ... and this engine makes a step's move at once (it knows the claim's run and index from its scan): the last one is taken
        // back in the array form — the claim returns to where it stood, with its new count (scheduler.go:598 sorts at the NEXT add)
        ClaimOrder<W, uint32_t*, false> t;
        t.key = order.key; t.ord = order.ord; t.pos = nullptr; t.n = n;
        const int q = (int)W::uniform((uint64_t)order.position(last_x));
        if (last_kind == 1) t.rotate_right(last_p, q); else t.rotate_left(q, n - 1);
      }
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
    c.cycles[0] = tc0; c.cycles[1] = tc1; c.cycles[2] = tc2; c.cycles[3] = tc3; c.cycles[4] = tc4; c.cycles[5] = tc5; c.cycles[6] = tc6; c.cycles[7] = tc7; c.cycles[8] = tc8; c.cycles[9] = tc9;
    if (W::leader()) *S.counters = c;
    W::sync();
  }

  KS_DEV static TopoRec load_rec(const KS_GLOBAL TopoRec* p) {   // two 16-byte loads
    TopoRec r;
    const KS_GLOBAL u32x4_alias* s = (const KS_GLOBAL u32x4_alias*)p;
    u32x4_alias* o = (u32x4_alias*)&r;
    o[0] = s[0]; o[1] = s[1];
    return r;
  }
  KS_DEV static void store_rec(KS_GLOBAL TopoRec* p, const TopoRec& r) {
    KS_GLOBAL u32x4_alias* d = (KS_GLOBAL u32x4_alias*)p;
    const u32x4_alias* o = (const u32x4_alias*)&r;
    d[0] = o[0]; d[1] = o[1];
  }

  // Solve — scheduler.go:440-519. The loop below keeps its own state in registers (claims, pods placed, counters; the classes of the
  // block's 64 pods one per lane) and reads LDS and HBM through pointers of its own: the engine object (`cold`, `order`, the
  // counters) is what the out-of-line paths work on, and the loop hands its registers over around every such call.
  KS_DEV void solve() {
    {
      const int why = (int)W::uniform((uint64_t)(uint32_t)cold.setup(true));
      if (why) { bail = why; finish(3, 0); return; }
      const int why2 = (int)W::uniform((uint64_t)(uint32_t)setup_topo());
      if (why2) { bail = why2; finish(3, 0); return; }
    }
    const int np = fast_uniform(Pk->n_pods);
    const int max_claims = fast_uniform(Sk->max_claims);
    KS_GLOBAL TopoRec* const rec = (KS_GLOBAL TopoRec*)fast_uniform(Tk->rec);
    KS_GLOBAL uint32_t* const ring = (KS_GLOBAL uint32_t*)fast_uniform(order.ring);
    KS_GLOBAL uint32_t* const ocnt = (KS_GLOBAL uint32_t*)fast_uniform(order.cnt);
    KS_GLOBAL uint32_t* const oslot = (KS_GLOBAL uint32_t*)fast_uniform(order.slot);
    KS_LDS RunTables* const RT = fast_uniform(order.T);
    KS_LDS TopoState* const S_ = fast_uniform(st);
    if (W::leader()) {
      S_->q_class = Fk->q_class; S_->q_claim = Fk->q_claim; S_->q_cnt = Fk->q_cnt; S_->tcls = Tk->cls; S_->fcls = Fk->cls; S_->hostseq = Fk->c_hostseq;
      S_->last[0] = 0; S_->last[1] = 0; S_->last[2] = 0; S_->last[3] = 0;
    }
    W::sync();
    const KS_LDS FastEnt* const ent = fast_uniform(cold.ent);
    const volatile int* const cancel = fast_uniform(Sk->cancel_flag);
    const long long max_steps = (long long)W::uniform((uint64_t)Sk->max_steps);
    const int kmax = fast_uniform(order.kmax);
    // the template a new claim comes from, when there is exactly one and it has no limits (the loop opens such claims itself)
    int fast_t = -1;
    {
      const uint32_t at = (uint32_t)fast_uniform((int)cold.active_templates);
      if (at && !(at & (at - 1)) && !Pk->tmpl_limit_mask[ctz64(at)]) fast_t = ctz64(at);
    }
    const uint64_t fast_tv = fast_t >= 0 ? W::uniform(Mp->tvmask[fast_t]) : 0ull;
    // ---- the loop's registers (the object's copies are written before / read after every out-of-line call) ----
    int n = 0, max_cnt = 1, n_claims = 0;            // order.n, order.max_cnt, cold.n_claims
    uint32_t host_seq = 0;                           // cold.host_seq
    uint64_t trk = W::uniform(track_fields);
    bool pending = false;                            // `order` holds a move for its next sort()
    unsigned long long ref = 0;
    uint32_t windows = 0, steps = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, t8 = 0;
    int status = 0;
    auto push = [&]() { order.n = n; order.max_cnt = max_cnt; cold.n_claims = n_claims; cold.host_seq = host_seq; track_fields = trk; last_kind = S_->last[0]; last_x = S_->last[1]; last_p = S_->last[2]; n_ref = ref; };
    auto set_last = [&](int kind, int x, int pos) { if (W::leader()) { S_->last[0] = kind; S_->last[1] = x; S_->last[2] = pos; } };
    auto pull = [&]() {
      n = fast_uniform(order.n); max_cnt = fast_uniform(order.max_cnt); n_claims = fast_uniform(cold.n_claims); host_seq = (uint32_t)fast_uniform((int)cold.host_seq);
      trk = W::uniform(track_fields); set_last(last_kind, last_x, last_p); ref = W::uniform(n_ref);
      pending = fast_uniform(order.defect_claim) >= 0;
    };
    // ---- the front of the order, in registers: the claims at positions 0 .. wn-1 with their records (lane = position) ----
    // Pods go to the emptiest claims (scheduler.go:598 sorts by pod count), so the first acceptor is almost always among the first
    // positions, and a step changes that front by ONE claim: the acceptor leaves its place (the claims behind it, up to its new place,
    // step one position to the left; it re-enters at the first position of the next run when that lies inside the window), a new claim
    // enters behind the claims with one pod. The loop keeps the window up to date with lane shuffles instead of reading the rings and
    // the records again: the common step issues NO load from HBM — only the stores of what it changed — and needs no fence; a fence
    // stands in front of the first load after stores (`dirty`). A claim that leaves beyond the window shrinks it (wn); it is read
    // again from the rings when it gets short, when the acceptor lies beyond it, and after every out-of-line path.
    LaneVar<uint32_t> wx, wc;          // claim, pod count
    LaneVar<uint64_t> wv, wh;          // requirement set, hostname counters
    LaneVar<int32_t> wq0, wq1, wq2, wq3;   // requests
    int wn = 0;
    bool wvalid = false, dirty = false;
    constexpr int kRefill = 24;
    W::each([&](int l) { wx.at(l) = 0; wc.at(l) = 0; wv.at(l) = 0; wh.at(l) = 0; wq0.at(l) = 0; wq1.at(l) = 0; wq2.at(l) = 0; wq3.at(l) = 0; });
    auto fence = [&]() { if (dirty) { W::sync(); dirty = false; } };
    // the window's lanes take the values of lanes srcf(lane); lane `ins` (if >= 0) takes the given claim instead
    auto wmove = [&](auto srcf, int ins, uint32_t ix, uint32_t ic, uint64_t iv, uint64_t ih, int32_t i0, int32_t i1, int32_t i2, int32_t i3) {
      LaneVar<uint32_t> nx, nc_; LaneVar<uint64_t> nv, nh; LaneVar<int32_t> n0, n1, n2, n3;
      W::each([&](int l) {
        const int src = srcf(l) & 63;
        nx.at(l) = wx.shuffle(l, src); nc_.at(l) = wc.shuffle(l, src); nv.at(l) = wv.shuffle(l, src); nh.at(l) = wh.shuffle(l, src);
        n0.at(l) = wq0.shuffle(l, src); n1.at(l) = wq1.shuffle(l, src); n2.at(l) = wq2.shuffle(l, src); n3.at(l) = wq3.shuffle(l, src);
      });
      W::each([&](int l) {
        const bool me = l == ins;
        wx.at(l) = me ? ix : nx.at(l); wc.at(l) = me ? ic : nc_.at(l); wv.at(l) = me ? iv : nv.at(l); wh.at(l) = me ? ih : nh.at(l);
        wq0.at(l) = me ? i0 : n0.at(l); wq1.at(l) = me ? i1 : n1.at(l); wq2.at(l) = me ? i2 : n2.at(l); wq3.at(l) = me ? i3 : n3.at(l);
      });
    };
    for (int base = 0; base < np && !status; base += 64) {
      const int bn = np - base < 64 ? np - base : 64;
      // the classes of the block's pods, one per lane: one gather per 64 pods, a handful of v_readlane per pod
      LaneVar<uint64_t> c_hlim, c_hinc, c_zsel, c_cvm, c_dm;
      LaneVar<int32_t> c_zg, c_s0, c_s1, c_s2, c_s3;
      LaneVar<uint32_t> c_zself, c_excl, c_tok, oclaim, ocntv;
      const KS_GLOBAL uint32_t* const gqcls = (const KS_GLOBAL uint32_t*)fast_uniform(S_->q_class);
      const KS_GLOBAL TopoClass* const gtc = (const KS_GLOBAL TopoClass*)fast_uniform(S_->tcls);
      const KS_GLOBAL FastSlot* const gcs = (const KS_GLOBAL FastSlot*)fast_uniform(S_->fcls);
      W::each([&](int l) {
        const int q = base + (l < bn ? l : bn - 1);
        const uint32_t k = gqcls[q] & ~kFastLastBit;
        const KS_GLOBAL u64_alias* tp = (const KS_GLOBAL u64_alias*)(gtc + k);
        const uint64_t w3 = tp[3], w4 = tp[4];
        c_hlim.at(l) = tp[0]; c_hinc.at(l) = tp[1]; c_zsel.at(l) = tp[2];
        c_zg.at(l) = (int32_t)(uint32_t)w3; c_zself.at(l) = (uint32_t)(w3 >> 32); c_excl.at(l) = (uint32_t)w4;   // (zg: two int16 in one word)
        const KS_GLOBAL u64_alias* sp = (const KS_GLOBAL u64_alias*)(gcs + k);
        const uint64_t v2 = sp[2], v3 = sp[3], v4 = sp[4];
        c_cvm.at(l) = sp[0]; c_dm.at(l) = sp[1];
        c_s0.at(l) = (int32_t)(uint32_t)v2; c_s1.at(l) = (int32_t)(uint32_t)(v2 >> 32); c_s2.at(l) = (int32_t)(uint32_t)v3; c_s3.at(l) = (int32_t)(uint32_t)(v3 >> 32);
        c_tok.at(l) = (uint32_t)v4;
        oclaim.at(l) = 0xFFFFFFFFu; ocntv.at(l) = 0;
      });
      if (cancel) {
        // > 0: ksolve_cancel / the deadline; < 0 (tests only, KSOLVE_TEST_CANCEL_AT): as if the cancel landed once -flag pods were placed
        const int cv = fast_uniform((int)W::poll_flag(cancel));
        if (cv > 0 || (cv < 0 && (long long)steps >= -(long long)cv)) { status = 2; break; }
      }
      int bi = 0;
      for (; bi < bn; ++bi) {
        if (max_steps >= 0 && (long long)steps >= max_steps) { status = 2; break; }
        steps++;
        unsigned long long tq = W::clock();
#define KS_TSEC(acc) { const unsigned long long tn_ = W::clock(); acc += tn_ - tq; tq = tn_; }
        if (pending) {   // scheduler.go:598: a move pdqsort makes the long way
          fence(); push(); sort_cold(); pull(); wvalid = false;
          if (fast_uniform((int)order.overflow)) { status = 1; break; }
        }
        KS_TSEC(t0)
        // ---- the pod's class ----
        KClass kc;
        kc.hlim = c_hlim.bcast(bi); kc.cvmask = c_cvm.bcast(bi); kc.dmask = c_dm.bcast(bi);
        kc.s0 = c_s0.bcast(bi); kc.s1 = c_s1.bcast(bi); kc.s2 = c_s2.bcast(bi); kc.s3 = c_s3.bcast(bi); kc.tmplok = c_tok.bcast(bi);
        const uint64_t hinc = c_hinc.bcast(bi), zsel = c_zsel.bcast(bi);
        const int zgw = c_zg.bcast(bi);
        const int zg0 = (int)(int16_t)(uint16_t)(uint32_t)zgw, zg1 = (int)(int16_t)(uint16_t)((uint32_t)zgw >> 16);
        const uint32_t zselfw = c_zself.bcast(bi);
        const uint32_t excl = c_excl.bcast(bi);
        KS_TSEC(t1)
        LaneVar<uint64_t> zkv;
        bool possible = true;
        const ZChoice zc = choose_domains(zg0, (zselfw & 1u) != 0, kc.cvmask, zkv, possible);
        if (!possible) { bail = 27; status = 3; break; }
        const bool two = zg1 >= 0;   // a second group on a dictionary key (rare): its pods take the out-of-line paths all the way
        KS_TSEC(t2)
        // ---- addToInflightNode (scheduler.go:658-692): the first claim of the order that accepts ----
        bool found = false;
        // the acceptor (wave-uniform): claim, position, its narrowed requirement set, hostname counters, requests, pod count
        uint32_t kx = 0, a_pos = 0, kcn = 0; uint64_t km = 0, kh = 0; int32_t k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        bool from_window = false;   // the acceptor is lane a_pos of the window registers
        const bool listed = excl != 0xFFu && fast_uniform((int)S_->track_field[excl & (kTopoTrack - 1)]) != 0xFF;
        if (listed) {
          // (usually nobody is on the list: every claim holds a member already — straight to addToNewNodeClaim)
          if (fast_uniform((int)S_->n_free[excl & (kTopoTrack - 1)]) > 0) {
            fence(); push();
            ZChoice zn; zn.vm = 0; zn.off = 0; zn.width = 0; zn.multi = false; zn.on = false; zn.clear = 0;
            const Accept A = two ? scan_two((int)excl, kc, zc, zkv, zg1, (zselfw & 2u) != 0) : scan_list((int)excl, kc, zc, zkv, zn, zkv);
            pull();
            if (fast_uniform(bail)) { status = 3; break; }
            if (fast_uniform(A.found)) {
              found = true; a_pos = (uint32_t)fast_uniform((int)A.pos); kx = (uint32_t)fast_uniform((int)A.x); kcn = (uint32_t)fast_uniform((int)A.cnt);
              km = W::uniform(A.m2); kh = W::uniform(A.hcnt); k0 = fast_uniform(A.q0); k1 = fast_uniform(A.q1); k2 = fast_uniform(A.q2); k3 = fast_uniform(A.q3);
            }
          }
        } else if (two) {
          fence(); push();
          const Accept A = scan_two(-1, kc, zc, zkv, zg1, (zselfw & 2u) != 0);
          pull();
          if (fast_uniform(bail)) { status = 3; break; }
          if (fast_uniform(A.found)) {
            found = true; a_pos = (uint32_t)fast_uniform((int)A.pos); kx = (uint32_t)fast_uniform((int)A.x); kcn = (uint32_t)fast_uniform((int)A.cnt);
            km = W::uniform(A.m2); kh = W::uniform(A.hcnt); k0 = fast_uniform(A.q0); k1 = fast_uniform(A.q1); k2 = fast_uniform(A.q2); k3 = fast_uniform(A.q3);
          }
        } else {
          const int wfull = n < 64 ? n : 64;
          if (!wvalid || (wn < kRefill && wn < wfull)) {
            // ---- (re)read the window: the claims at positions 0 .. 63 from the rings, then their records ----
            fence(); push();
            Win w = read_window();
            pull();
            if (fast_uniform(bail)) { status = 3; break; }
            W::each([&](int l) { wx.at(l) = w.x.at(l); wc.at(l) = w.c.at(l); wv.at(l) = w.v.at(l); wh.at(l) = w.h.at(l); wq0.at(l) = w.q0.at(l); wq1.at(l) = w.q1.at(l); wq2.at(l) = w.q2.at(l); wq3.at(l) = w.q3.at(l); });
            wn = fast_uniform(w.n); wvalid = true;
            windows++;
          }
          KS_TSEC(t3)
          {
            // ---- CanAdd on the window's claims, from registers ----
            LaneVar<uint64_t> m2v;
            uint64_t okm = 0, und = 0;
            const int lim = wn;
            W::ballot2([&](int l) {
              uint64_t m2;
              const int v = lane_test<false>(wv.at(l), wq0.at(l), wq1.at(l), wq2.at(l), wq3.at(l), wh.at(l), kc, zc, zkv, zc, zkv, ent, m2);
              m2v.at(l) = m2;
              return l < lim ? v : 0;
            }, okm, und);
            KS_TSEC(t4)
            const uint64_t before = okm ? (und & ((1ull << ctz64(okm)) - 1)) : und;
            if (before) { fence(); push(); okm |= resolve(before, m2v, wq0, wq1, wq2, wq3, kc.s0, kc.s1, kc.s2, kc.s3); pull(); if (fast_uniform(bail)) { status = 3; break; } }
            if (okm) {
              const int a = ctz64(okm);
              found = true; from_window = true; a_pos = (uint32_t)a;
              kx = wx.bcast(a); km = m2v.bcast(a); kh = wh.bcast(a); kcn = wc.bcast(a);
              k0 = wq0.bcast(a); k1 = wq1.bcast(a); k2 = wq2.bcast(a); k3 = wq3.bcast(a);
            }
          }
          if (!found && wn < n) {
            fence(); push();
            ZChoice zn; zn.vm = 0; zn.off = 0; zn.width = 0; zn.multi = false; zn.on = false; zn.clear = 0;
            const Accept A = scan_order(kc, zc, zkv, zn, zkv);
            pull();
            if (fast_uniform(bail)) { status = 3; break; }
            if (fast_uniform(A.found)) {
              found = true; a_pos = (uint32_t)fast_uniform((int)A.pos); kx = (uint32_t)fast_uniform((int)A.x); kcn = (uint32_t)fast_uniform((int)A.cnt);
              km = W::uniform(A.m2); kh = W::uniform(A.hcnt); k0 = fast_uniform(A.q0); k1 = fast_uniform(A.q1); k2 = fast_uniform(A.q2); k3 = fast_uniform(A.q3);
            }
          }
        }
        KS_TSEC(t5)
        if (!found) {
          // ---- addToNewNodeClaim (scheduler.go:695-790) ----
          bool made = false;
          if (!two && fast_t >= 0 && n >= 50 && n_claims < max_claims && max_cnt + 2 < kRunMaxCount) {
            // one template without limits, the order past pdqsort's small-array paths: the claim is opened here. NewNodeClaim draws a
            // hostname number, CanAdd runs on the template's requirement set with nothing requested and every hostname counter zero
            uint64_t m2 = 0;
            const int v = fast_uniform(lane_test<false>(fast_tv, 0, 0, 0, 0, 0ull, kc, zc, zkv, zc, zkv, ent, m2));
            m2 = W::uniform(m2);
            if (v == 1) {
              ref += (unsigned long long)n + 1;
              host_seq++;
              const int c = n_claims++;
              TopoRec nrq;
              nrq.vmask = m2; nrq.req[0] = kc.s0; nrq.req[1] = kc.s1; nrq.req[2] = kc.s2; nrq.req[3] = kc.s3; nrq.hcnt = host_add(0, hinc);
              // the new claim: behind the last claim with one pod — the end of run 1; every later run starts one position further right
              const RunEnt e1 = lds_get16(&RT->e[1]);
              const uint32_t m1 = (1u << RT->log2cap[1]) - 1u;
              const uint32_t s1 = ((uint32_t)fast_uniform((int)e1.head) + (uint32_t)fast_uniform((int)e1.size)) & (uint32_t)fast_uniform((int)m1);
              const uint32_t o1 = (uint32_t)fast_uniform((int)e1.off);
              if (W::leader()) {
                store_rec(rec + c, nrq); ((KS_GLOBAL uint32_t*)S_->hostseq)[c] = host_seq;
                ring[o1 + s1] = (uint32_t)c; oslot[c] = s1; ocnt[c] = 1u;
                RT->e[1].size = e1.size + 1;
              }
              const int top = max_cnt + 1;
              W::each([&](int l) { for (int kk = 2 + l; kk <= top; kk += 64) RT->e[kk].prefix += 1; });
              n++;
              record_zonal(zsel, m2);
              if (trk) {
                // the lists of claims without a member of an anti-affinity group: the new claim joins those whose counter it leaves at zero
                const uint64_t zero_now = ~(nrq.hcnt | (nrq.hcnt >> 1) | (nrq.hcnt >> 2)) & kTopoOnes & trk;
                if (zero_now) { push(); lists_add(nrq.hcnt, (uint32_t)c); pull(); }
              }
              oclaim.set(bi, (uint32_t)c); ocntv.set(bi, 0u);
              set_last(2, c, 0);
              dirty = true;
              if (wvalid) {
                // the window: the new claim stands at position b = the claims with one pod in front of it; the claims from b on step right
                const int b = fast_uniform((int)e1.size);
                if (b <= wn) {
                  wmove([&](int l) { return l > b ? l - 1 : l; }, b, (uint32_t)c, 1u, m2, nrq.hcnt, kc.s0, kc.s1, kc.s2, kc.s3);
                  if (wn < 64) wn++;
                }
              }
              made = true;
            }
          }
          if (!made) {
            push();
            TopoClass tcc; tcc.hlim = kc.hlim; tcc.hinc = hinc; tcc.zsel = zsel; tcc.zg[0] = (int16_t)zg0; tcc.zg[1] = (int16_t)zg1; tcc.zself = zselfw; tcc.excl = excl; tcc.pad = 0;
            FastSlot csc; csc.cvmask = kc.cvmask; csc.dmask = kc.dmask; csc.size[0] = kc.s0; csc.size[1] = kc.s1; csc.size[2] = kc.s2; csc.size[3] = kc.s3; csc.tmplok = kc.tmplok; csc.kdef = 0;
            fence();
            const int c = fast_uniform(new_claim(tcc, csc));
            pull(); wvalid = false;
            if (c < 0) { status = fast_uniform(bail) < 0 ? 1 : 3; break; }
            oclaim.set(bi, (uint32_t)c); ocntv.set(bi, 0u);
          }
          KS_TSEC(t6)
          continue;
        }
        // ---- NodeClaim.Add (nodeclaim.go:247-263) ----
        ref += (unsigned long long)a_pos + 1;
        TopoRec nrq;
        nrq.vmask = km;
        nrq.req[0] = k0 + kc.s0; nrq.req[1] = k1 + kc.s1; nrq.req[2] = k2 + kc.s2; nrq.req[3] = k3 + kc.s3;
        nrq.hcnt = host_add(kh, hinc);
        if (W::leader()) store_rec(rec + kx, nrq);
        dirty = true;
        oclaim.set(bi, kx); ocntv.set(bi, kcn);
        record_zonal(zsel, km);
        // the anti-affinity lists: the claim leaves those whose counter this pod takes from zero
        {
          const uint64_t zero_before = ~(kh | (kh >> 1) | (kh >> 2)) & kTopoOnes;
          const uint64_t leaving = hinc & zero_before & trk;
          if (leaving) { push(); lists_remove(leaving, kx); pull(); }   // (LDS only)
        }
        KS_TSEC(t7)
        // The sort.Slice of the NEXT add (scheduler.go:598) moves this claim behind the claims with fewer pods. When that is pdqsort's
        // single stable move — the claim leaves its run and becomes the first of the next one (run_order.h) — it is made now: run and
        // index are known from the scan, no load stands in front of it.
        {
          const int k = (int)kcn;
          const unsigned q4 = (unsigned)n >> 2;
          const bool single = n <= 12 || (n >= 50 && !(a_pos - (q4 - 1) <= 2u || a_pos - (2 * q4 - 1) <= 2u || a_pos - (3 * q4 - 1) <= 2u));
          bool moved = false;
          if (from_window && single && k + 3 < kRunMaxCount && k + 2 < kmax) {
            const RunEnt ea = lds_get16(&RT->e[k]), eb = lds_get16(&RT->e[k + 1]);
            const uint32_t ma = (1u << RT->log2cap[k]) - 1u, mb = (1u << RT->log2cap[k + 1]) - 1u;
            const uint32_t h = (uint32_t)fast_uniform((int)ea.head), sz = (uint32_t)fast_uniform((int)ea.size), oa = (uint32_t)fast_uniform((int)ea.off), m = (uint32_t)fast_uniform((int)ma);
            const uint32_t idx = a_pos - (uint32_t)fast_uniform((int)ea.prefix);
            uint32_t hb = (uint32_t)fast_uniform((int)eb.head), sb = (uint32_t)fast_uniform((int)eb.size);
            const uint32_t pb = (uint32_t)fast_uniform((int)eb.prefix), ob = (uint32_t)fast_uniform((int)eb.off), mbu = (uint32_t)fast_uniform((int)mb);
            if (k + 1 > max_cnt) {
              // the first claim with k + 1 pods: run k + 1 starts out empty, and every claim lies in front of run k + 2 (RunOrder::grow_to)
              hb = 0; sb = 0;
              if (W::leader()) RT->e[k + 2].prefix = (uint32_t)n;
              max_cnt = k + 1;
            }
            uint32_t new_head = h;
            if (idx == 0) new_head = (h + 1) & m;
            else if (idx + 1 < sz) {
              // the claims in front of it inside its run are window lanes: they step one ring slot towards the hole, from registers
              const uint32_t pk = a_pos - idx;
              W::each([&](int l) { if ((uint32_t)l >= pk && (uint32_t)l < a_pos) { const uint32_t s2 = (h + ((uint32_t)l - pk) + 1u) & m; ring[oa + s2] = wx.at(l); oslot[wx.at(l)] = s2; } });
              new_head = (h + 1) & m;
            }
            const uint32_t h1 = (hb - 1u) & mbu;
            if (W::leader()) {
              ring[ob + h1] = kx; oslot[kx] = h1; ocnt[kx] = (uint32_t)(k + 1);
              RT->e[k].head = new_head; RT->e[k].size = sz - 1;
              RT->e[k + 1].head = h1; RT->e[k + 1].size = sb + 1; RT->e[k + 1].prefix = pb - 1;
            }
            set_last(1, (int)kx, (int)a_pos);
            moved = true;
            dirty = true;
            // the window: the claims behind the acceptor, up to its new place q, step one position to the left; it re-enters at q
            // when q lies inside the window, otherwise the window is one claim shorter
            const int a = (int)a_pos, q = (int)pb - 1;
            const bool inside = q < wn;
            wmove([&](int l) { return (l >= a && l < q) ? l + 1 : l; }, inside ? q : -1, kx, (uint32_t)(k + 1), nrq.vmask, nrq.hcnt, nrq.req[0], nrq.req[1], nrq.req[2], nrq.req[3]);
            if (!inside) wn--;
          }
          if (!moved) {
            fence(); push(); move_cold((int)kx, k, a_pos); pull(); wvalid = false;
            if (fast_uniform((int)order.overflow)) { status = 1; break; }
          }
        }
        KS_TSEC(t8)
      }
#undef KS_TSEC
      // the block's results, in queue order (ksolve_fast_scatter puts them under the pod indices)
      {
        const int dn = bi < bn ? bi : bn;
        KS_GLOBAL uint32_t* const gqclaim = (KS_GLOBAL uint32_t*)fast_uniform(S_->q_claim);
        KS_GLOBAL uint32_t* const gqcnt = (KS_GLOBAL uint32_t*)fast_uniform(S_->q_cnt);
        W::each([&](int l) { if (l < dn) { gqclaim[base + l] = oclaim.at(l); gqcnt[base + l] = ocntv.at(l); } });
      }
    }
    push();
    n_tests = (unsigned long long)windows * 64; n_windows = windows; n_listed = 0;
    tc0 = t0; tc1 = t1; tc2 = t2; tc3 = t3; tc4 = t4; tc5 = t5; tc6 = t6; tc7 = t7; tc8 = t8;
    W::sync();
    finish(status, steps);
  }
};

}  // namespace ks
