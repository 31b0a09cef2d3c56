// engine.h — the pack engine: the reference's Scheduler.Solve() loop (scheduler.go:440-519) for one scheduling
// problem, executed by ONE wavefront.
//
// Why one wave: first-fit-decreasing is a serial chain — every commit changes the claim it lands on, the claim order
// (scheduler.go:598) and the NodePool limits the next pod sees — so control is wave-uniform scalar code. The 64 lanes
// are the vector unit for what IS parallel inside a step:
//   * instance-type filtering (filterInstanceTypesByRequirements, nodeclaim.go:541-618): one lane per instance type,
//     requirement compatibility from precomputed per-(key,value) instance-type bitmasks, resource fit and offering
//     availability per lane, __ballot -> one u64 word of the surviving InstanceTypeOptions mask;
//   * first-fit selection over the in-flight claims in the reference's order with "lowest index wins"
//     (scheduler.go:667-686): ballot + ffs over candidate lists;
//   * the claim re-ordering (pdq_emul.h) and bulk state moves.
// Independent problems (NodePool components, consolidation probes) run as independent waves on other CUs.
//
// Exact pruning: dead[class][claim] caches "CanAdd(claim, pod of this class) failed". Between two changes of a
// claim's requirement set that verdict is monotone (requests only grow, InstanceTypeOptions only shrink), so the bit
// stays valid; when a commit changes the claim's requirements the claim's column is cleared for every class.
#pragma once
#include "ksp.h"
#include "pdq_emul.h"

namespace ks {

enum {
  E_OK = 0, E_TAINTS = 1, E_INCOMPATIBLE = 2, E_TOPOLOGY = 3, E_INSTANCE_TYPES = 4, E_RESOURCES = 5, E_NO_TEMPLATES = 6,
  E_LIMITS = 7, E_RESERVED = 8, E_EXISTING = 9, E_MIN_VALUES = 10
};

struct Scratch {
  ReqBuf merged;
  uint64_t cm[kMaxItWords];    // instance types compatible with `merged`
  uint64_t its[kMaxItWords];   // surviving InstanceTypeOptions
  uint64_t lim[kMaxItWords];   // instance types within NodePool limits
  uint64_t cand[64];           // candidate list: (position << 32 | claim)
  int64_t total[kMaxRes];
  int64_t head[kMaxRes];
};

template <class W>
struct Engine {
  const ProblemView& P;
  Workspace& S;
  Scratch& sc;
  ClaimOrder<W> order;
  int n_claims = 0;
  uint32_t host_seq = 0;
  uint32_t active_templates = 0;
  int last_err = 0, last_diag = 0;
  Counters ctr{};

  KS_FN Engine(const ProblemView& p, Workspace& s, Scratch& scratch) : P(p), S(s), sc(scratch) {
    order.key = s.o_key; order.ord = s.o_ord; order.pos = s.o_pos;
  }

  // ------------------------------------------------------------------------------------------------------------
  // Instance types compatible with a requirement set: InstanceType.Requirements.Intersects(reqs) == nil for every
  // instance type at once (nodeclaim.go:620-622, requirements.go:254-274), one u64 word per 64 instance types.
  KS_FN void compat_mask(const ReqBuf& m) {
    const Dict& d = P.dict;
    const ProblemView& Pv = P;
    uint64_t* cm = sc.cm;
    const int iw = P.it_words;
    W::for_n(iw, [&](int w) {
      uint64_t acc = ~0ull;
      uint32_t keys = m.defined;
      while (keys) {
        int k = __builtin_ctz(keys);
        keys &= keys - 1;
        uint32_t w0 = d.key_word_off[k], w1 = d.key_word_off[k + 1];
        bool comp = bit(m.complement, k);
        bool hg = bit(m.has_gte, k), hl = bit(m.has_lte, k);
        if (k == d.key_it) {
          // the instance-type key's dictionary IS the instance-type list and every type requires In [own name]
          uint64_t mm = m.mask[w0 + w];
          uint64_t v = comp ? inbounds_word(d, w0 + w, ~mm, hg, m.gte[k], hl, m.lte[k]) : mm;
          acc &= v;
          continue;
        }
        uint64_t r = Pv.key_undef[(size_t)k * iw + w];
        bool nonempty = false;
        if (!comp) {
          for (uint32_t x = w0; x < w1; ++x) {
            uint64_t bits = m.mask[x];
            if (bits) nonempty = true;
            while (bits) { int b = ctz64(bits); bits &= bits - 1; r |= Pv.kv_has[((size_t)x * 64 + b) * iw + w]; }
          }
          if (!nonempty) r |= Pv.key_neg[(size_t)k * iw + w];  // DoesNotExist vs {NotIn, DoesNotExist}: requirements.go:260-265
        } else {
          r |= Pv.key_compl[(size_t)k * iw + w];               // two complements always intersect: requirement.go:226-228
          for (uint32_t x = w0; x < w1; ++x) {
            if (m.mask[x]) nonempty = true;
            uint64_t bits = inbounds_word(d, x, ~m.mask[x] & d.value_valid[x], hg, m.gte[k], hl, m.lte[k]);
            while (bits) { int b = ctz64(bits); bits &= bits - 1; r |= Pv.kv_has[((size_t)x * 64 + b) * iw + w]; }
          }
          if (nonempty) r |= Pv.key_neg[(size_t)k * iw + w];   // NotIn vs {NotIn, DoesNotExist}
        }
        acc &= r;
      }
      cm[w] = acc;
    });
  }

  // zone x capacity-type cells an offering may sit in to be compatible with `m` (types.go:553-570:
  // reqs.IsCompatible(offering.Requirements, AllowUndefinedWellKnownLabels); offerings carry single In values).
  KS_FN uint64_t offering_cells(const ReqBuf& m) {
    const Dict& d = P.dict;
    ReqRef r = m.ref();
    uint32_t zones = 0, cts = 0;
    if (d.key_zone >= 0 && bit(m.defined, d.key_zone)) {
      for (int z = 0; z < P.n_zones; ++z) if (req_has(d, r, d.key_zone, d.key_word_off[d.key_zone], z)) zones |= 1u << z;
    } else zones = (1u << P.n_zones) - 1;
    if (d.key_ct >= 0 && bit(m.defined, d.key_ct)) {
      for (int c = 0; c < P.n_cts; ++c) if (req_has(d, r, d.key_ct, d.key_word_off[d.key_ct], c)) cts |= 1u << c;
    } else cts = (1u << P.n_cts) - 1;
    uint64_t cells = 0;
    while (zones) { int z = __builtin_ctz(zones); zones &= zones - 1; cells |= (uint64_t)cts << (z * 4); }
    return cells;
  }

  // filterInstanceTypesByRequirements for one candidate bin: its' = its ∩ compatible ∩ fits ∩ hasOffering.
  // Returns whether any instance type survives; fills sc.its. diag (InstanceTypeFilterError flags) only if asked.
  KS_FN bool filter_instance_types(const uint64_t* bin_its, const int64_t* total, bool want_diag) {
    compat_mask(sc.merged);
    uint64_t cells = offering_cells(sc.merged);
    const ProblemView& Pv = P;
    const int nr = P.n_res, n_its = P.n_its;
    uint64_t any = 0;
    bool d_req = false, d_fit = false, d_off = false, d_ro = false, d_fo = false;
    for (int w = 0; w < P.it_words; ++w) {
      uint64_t in = bin_its[w];
      if (!in) { W::store(&sc.its[w], (uint64_t)0); continue; }
      uint64_t cm = sc.cm[w];
      if (!want_diag && !(in & cm)) { W::store(&sc.its[w], (uint64_t)0); continue; }
      uint64_t aok = Pv.it_alloc_ok[w];
      uint64_t fit = W::ballot([&](int l) {
        int it = w * 64 + l;
        if (it >= n_its || !((in >> l) & 1)) return false;
        bool f = (aok >> l) & 1;
        for (int r = 0; r < nr; ++r) f = f && total[r] <= Pv.it_alloc[(size_t)r * n_its + it];
        return f;
      });
      uint64_t off = W::ballot([&](int l) {
        int it = w * 64 + l;
        if (it >= n_its || !((in >> l) & 1)) return false;
        return (Pv.it_off_avail[it] & cells) != 0;
      });
      ctr.it_evaluations += popc64(in);
      uint64_t itfits = fit & off;  // fits() reports itFits only together with a compatible offering (nodeclaim.go:624-638)
      uint64_t keep = in & cm & itfits;
      if (want_diag) {
        d_req |= (in & cm) != 0; d_fit |= itfits != 0; d_off |= off != 0;
        d_ro |= (in & cm & off & ~itfits) != 0; d_fo |= (itfits & ~cm) != 0;
      }
      W::store(&sc.its[w], keep);
      any |= keep;
    }
    W::sync();
    if (want_diag) last_diag = (d_req ? 1 : 0) | (d_fit ? 2 : 0) | (d_off ? 4 : 0) | (d_ro ? 16 : 0) | (d_fo ? 32 : 0);
    return any != 0;
  }

  // ------------------------------------------------------------------------------------------------------------
  // NodeClaim.CanAdd (nodeclaim.go:124-242) for a pod of class k on a bin described by (reqs, its, total, taints).
  // On success sc.merged / sc.its / sc.total hold the updated requirements, instance types and requests.
  KS_FN int can_add(int k, const ReqRef& bin_reqs, const uint64_t* bin_its, const int64_t* bin_total, const int64_t* bin_head,
                    uint64_t bin_taints, bool want_diag, bool* reqs_changed) {
    const Dict& d = P.dict;
    ctr.bin_evaluations++;
    if (bin_taints & ~P.cls_tolerates[k]) return E_TAINTS;                       // Taints.ToleratesPod — nodeclaim.go:126
    const int64_t* req = P.cls_requests + (size_t)k * P.n_res;
    if (bin_head) for (int r = 0; r < P.n_res; ++r) if (req[r] > bin_head[r]) return E_INSTANCE_TYPES;  // no remaining type can hold it
    ReqRef q = P.cls_reqs.at(d, k);
    int hn = d.key_hostname;
    if (hn >= 0 && bit(q.defined, hn)) {
      // the claim's own hostname requirement is In [hostname-placeholder-N] (nodeclaim.go:97), a value outside every
      // dictionary: only an unbounded complement (NotIn / Exists) on the pod side intersects it.
      if (!bit(q.complement, hn) || bit(q.has_gte | q.has_lte, hn)) return E_INCOMPATIBLE;
      q.defined &= ~(1u << hn);
    }
    if (reqs_compatible(d, bin_reqs, q, true) != COMPAT_OK) return E_INCOMPATIBLE;  // nodeclaim.go:133
    ctr.full_evaluations++;
    reqbuf_load(d, sc.merged, bin_reqs);
    bool changed = reqbuf_add(d, sc.merged, q);                                      // nodeclaim.go:136
    if (reqs_changed) *reqs_changed = changed;
    for (int r = 0; r < P.n_res; ++r) W::store(&sc.total[r], bin_total[r] + req[r]);  // resources.Merge — nodeclaim.go:211
    W::sync();
    if (!filter_instance_types(bin_its, sc.total, want_diag)) return E_INSTANCE_TYPES;  // nodeclaim.go:213
    return E_OK;
  }

  // ---- claim state ------------------------------------------------------------------------------------------
  KS_FN void write_claim(int c, int tmpl, bool fresh) {
    const Dict& d = P.dict;
    const int nr = P.n_res, iw = P.it_words;
    if (fresh) { W::store(&S.c_tmpl[c], (int32_t)tmpl); W::store(&S.c_host_seq[c], host_seq); W::store(&S.c_relaxed[c], (uint8_t)0); }
    uint64_t* cits = S.c_its + (size_t)c * iw;
    const uint64_t* sits = sc.its;
    W::for_n(iw, [&](int w) { cits[w] = sits[w]; });
    uint64_t* cm = S.c_reqs.mask + (size_t)c * d.req_words;
    const ReqBuf& m = sc.merged;
    W::for_n(d.req_words, [&](int w) { cm[w] = m.mask[w]; });
    W::store(&S.c_reqs.defined[c], m.defined); W::store(&S.c_reqs.complement[c], m.complement);
    W::store(&S.c_reqs.has_gte[c], m.has_gte); W::store(&S.c_reqs.has_lte[c], m.has_lte);
    int64_t* cg = S.c_reqs.gte + (size_t)c * d.n_keys; int64_t* cl = S.c_reqs.lte + (size_t)c * d.n_keys;
    int32_t* cv = S.c_reqs.minv + (size_t)c * d.n_keys;
    W::for_n(d.n_keys, [&](int k) { cg[k] = m.gte[k]; cl[k] = m.lte[k]; cv[k] = m.minv[k]; });
    // headroom = max allocatable over the surviving instance types - total
    int64_t* tot = S.c_total + (size_t)c * nr;
    int64_t* head = S.c_head + (size_t)c * nr;
    const ProblemView& Pv = P;
    const int n_its = P.n_its;
    bool is_closed = false;
    for (int r = 0; r < nr; ++r) {
      int64_t mx = W::reduce_max_i64(n_its, [&](int it) { return ((sits[it >> 6] >> (it & 63)) & 1) ? Pv.it_alloc[(size_t)r * n_its + it] : INT64_MIN; });
      int64_t h = mx - sc.total[r];
      W::store(&tot[r], sc.total[r]);
      W::store(&head[r], h);
      if (P.min_request[r] > 0 && h < P.min_request[r]) is_closed = true;
    }
    if (is_closed) W::store(&S.closed[c >> 6], (uint64_t)(S.closed[c >> 6] | (1ull << (c & 63))));
    W::sync();
  }
  KS_FN void reset_column(int c) {
    ctr.column_resets++;
    uint64_t* dead = S.dead;
    const int cw = S.claim_words;
    const uint64_t clr = ~(1ull << (c & 63));
    const int word = c >> 6;
    W::for_n(P.n_classes, [&](int k) { dead[(size_t)k * cw + word] &= clr; });
  }
  KS_FN void mark_dead(int k, int c) {
    uint64_t* p = &S.dead[(size_t)k * S.claim_words + (c >> 6)];
    W::store(p, (uint64_t)(*p | (1ull << (c & 63))));
    W::sync();
  }
  KS_FN void commit_pod(int pod, int claim, uint32_t slot) {
    W::store(&S.assign[pod], (int32_t)claim);
    W::store(&S.slot[pod], slot);
  }

  // ---- in-flight scan: addToInflightNode (scheduler.go:658-692) ---------------------------------------------
  // returns E_OK when the pod was committed to some claim
  KS_FN int try_claim(int k, int c, int pod) {
    const Dict& d = P.dict;
    ReqRef cr = S.c_reqs.at(d, c);
    bool changed = false;
    int rc = can_add(k, cr, S.c_its + (size_t)c * P.it_words, S.c_total + (size_t)c * P.n_res, S.c_head + (size_t)c * P.n_res,
                     P.tmpl_taints[S.c_tmpl[c]], false, &changed);
    if (rc != E_OK) { mark_dead(k, c); return rc; }
    uint32_t np = S.c_npods[c];
    ctr.ref_bin_evaluations += (unsigned long long)order.pos[c] + 1;   // the reference walked every claim up to this position
    write_claim(c, S.c_tmpl[c], false);
    W::store(&S.c_npods[c], np + 1);
    order.increment(c);
    if (changed) reset_column(c);
    commit_pod(pod, c, np);
    return E_OK;
  }
  KS_FN bool scan_inflight(int k, int pod) {
    if (n_claims == 0) return false;
    const int words = (n_claims + 63) >> 6;
    const uint64_t* drow = S.dead + (size_t)k * S.claim_words;
    const uint64_t* closed = S.closed;
    const int nc = n_claims;
    // Steady state: only a handful of claims are not yet known infeasible for this class. Find the words of the
    // class's dead row that still have live bits (one ballot per 64 words), gather those claims and probe them in
    // position order (lowest position wins, scheduler.go:673-676).
    int ncand = 0;
    bool overflow = false;
    for (int w0 = 0; w0 < words && !overflow; w0 += 64) {
      int wn = words - w0 < 64 ? words - w0 : 64;
      uint64_t any = W::ballot([&](int l) {
        if (l >= wn) return false;
        int w = w0 + l;
        uint64_t valid = (w == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
        return (~drow[w] & ~closed[w] & valid) != 0;
      });
      while (any && !overflow) {
        int w = w0 + ctz64(any);
        any &= any - 1;
        uint64_t valid = (w == words - 1 && (nc & 63)) ? ((1ull << (nc & 63)) - 1) : ~0ull;
        uint64_t a = ~drow[w] & ~closed[w] & valid;
        while (a) {
          int b = ctz64(a); a &= a - 1;
          if (ncand >= 64) { overflow = true; break; }
          int c = w * 64 + b;
          W::store(&sc.cand[ncand], ((uint64_t)order.pos[c] << 32) | (uint32_t)c);
          ncand++;
        }
      }
    }
    W::sync();
    if (!overflow) {
      if (ncand == 0) return false;
      for (;;) {
        const uint64_t* cand = sc.cand;
        uint64_t best = W::reduce_min(ncand, [&](int i) { return cand[i]; });
        if (best == ~0ull) return false;
        int c = (int)(uint32_t)best;
        if (try_claim(k, c, pod) == E_OK) return true;
        for (int i = 0; i < ncand; ++i) if (sc.cand[i] == best) W::store(&sc.cand[i], (uint64_t)~0ull);
        W::sync();
      }
    }
    // Many live claims (a class seen for the first time, or very many open bins): walk the claims in the reference's
    // order, 64 positions per step, ballot the live ones and probe them lowest position first.
    ctr.walk_scans++;
    for (int base = 0; base < nc; base += 64) {
      const uint32_t* ord = order.ord;
      uint64_t m = W::ballot([&](int l) {
        int i = base + l;
        if (i >= nc) return false;
        uint32_t c = ord[i];
        return !(((drow[c >> 6] | closed[c >> 6]) >> (c & 63)) & 1);
      });
      while (m) {
        int l = ctz64(m); m &= m - 1;
        int c = (int)order.ord[base + l];
        if (try_claim(k, c, pod) == E_OK) return true;
      }
    }
    return false;
  }

  // ---- new claim: addToNewNodeClaim (scheduler.go:695-790) --------------------------------------------------
  KS_FN int add_to_new_claim(int k, int pod) {
    const Dict& d = P.dict;
    const int nr = P.n_res, iw = P.it_words, n_its = P.n_its;
    int first_err = 0, first_diag = 0;
    ctr.ref_bin_evaluations += (unsigned long long)n_claims;  // the reference tried every in-flight claim before coming here
    for (int t = 0; t < P.n_templates; ++t) {
      if (!((active_templates >> t) & 1)) continue;
      const uint64_t* its = S.t_its + (size_t)t * iw;
      uint32_t lm = P.tmpl_limit_mask[t];
      if (lm) {
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        if (((lm >> nr) & 1) && rem[nr] == 0) { if (!first_err) first_err = E_LIMITS; continue; }   // node limit — scheduler.go:711-715
        // filterByRemainingResources — scheduler.go:1069-1085
        uint64_t any = 0;
        const ProblemView& Pv = P;
        for (int w = 0; w < iw; ++w) {
          uint64_t in = its[w];
          uint64_t ok = in ? W::ballot([&](int l) {
            int it = w * 64 + l;
            if (it >= n_its || !((in >> l) & 1)) return false;
            bool v = true;
            for (int r = 0; r < nr; ++r) if ((lm >> r) & 1) v = v && Pv.it_cap[(size_t)r * n_its + it] <= rem[r];
            if ((lm >> nr) & 1) v = v && 0 <= rem[nr];  // instance types carry no "nodes" capacity (scheduler.go:1076)
            return v;
          }) : 0;
          W::store(&sc.lim[w], ok);
          any |= ok;
        }
        W::sync();
        if (!any) { if (!first_err) first_err = E_LIMITS; continue; }
        its = sc.lim;
      }
      host_seq++;  // NewNodeClaim draws a hostname-placeholder number for every attempt (nodeclaim.go:93)
      int64_t zero[kMaxRes];
      for (int r = 0; r < nr; ++r) zero[r] = 0;
      bool changed;
      ctr.ref_bin_evaluations++;
      int rc = can_add(k, P.tmpl_reqs.at(d, t), its, zero, nullptr, P.tmpl_taints[t], first_err == 0, &changed);
      if (rc != E_OK) { if (!first_err) { first_err = rc; first_diag = rc == E_INSTANCE_TYPES ? last_diag : 0; } continue; }
      if (n_claims >= S.max_claims) { W::store(S.status_out, 1); return -1; }
      int c = n_claims++;
      write_claim(c, t, true);
      W::store(&S.c_npods[c], 1u);
      order.append(c);
      if (lm) {
        // subtractMax — scheduler.go:1049-1066 : remaining -= max capacity over the claim's instance types
        int64_t* rem = S.t_remaining + (size_t)t * (nr + 1);
        const uint64_t* sits = sc.its;
        const ProblemView& Pv = P;
        for (int r = 0; r < nr; ++r) if ((lm >> r) & 1) {
          int64_t mx = W::reduce_max_i64(n_its, [&](int it) { return ((sits[it >> 6] >> (it & 63)) & 1) ? Pv.it_cap[(size_t)r * n_its + it] : INT64_MIN; });
          W::store(&rem[r], rem[r] - mx);
        }
        W::sync();
      }
      commit_pod(pod, c, 0);
      return E_OK;
    }
    last_diag = first_diag;
    return first_err ? first_err : E_NO_TEMPLATES;
  }

  // add — scheduler.go:582-612
  KS_FN int add(int row, int pod) {
    int k = (int)P.row_class[row];
    ctr.sorts++;
    order.sort();                                      // scheduler.go:598
    if (scan_inflight(k, pod)) return E_OK;            // scheduler.go:601
    if (active_templates == 0) { last_diag = 0; return E_NO_TEMPLATES; }   // scheduler.go:604-606
    return add_to_new_claim(k, pod);                   // scheduler.go:607
  }
  // trySchedule — scheduler.go:521-552 ; the relaxation ladder (preferences.go:38-57) is precomputed as row chain
  KS_FN int try_schedule(int pod) {
    int row = pod;
    for (;;) {
      int rc = add(row, pod);
      if (rc == E_OK || rc < 0) return rc;
      if (rc == E_RESERVED) return rc;
      int nxt = P.row_next[row];
      if (nxt < 0) return rc;
      row = nxt;
      ctr.relaxations++;
    }
  }

  // NewScheduler's per-template prefilter (scheduler.go:156-171): instance types compatible with the template's own
  // requirements, with non-negative allocatable and a compatible available offering.
  KS_FN void prefilter_templates() {
    const Dict& d = P.dict;
    active_templates = 0;
    int64_t zero[kMaxRes];
    for (int r = 0; r < P.n_res; ++r) zero[r] = 0;
    for (int t = 0; t < P.n_templates; ++t) {
      reqbuf_load(d, sc.merged, P.tmpl_reqs.at(d, t));
      bool any = filter_instance_types(P.tmpl_its + (size_t)t * P.it_words, zero, false);
      uint64_t* dst = S.t_its + (size_t)t * P.it_words;
      const uint64_t* sits = sc.its;
      W::for_n(P.it_words, [&](int w) { dst[w] = sits[w]; });
      if (any) active_templates |= 1u << t;
      int64_t* rem = S.t_remaining + (size_t)t * (P.n_res + 1);
      const int64_t* lim = P.tmpl_limits + (size_t)t * (P.n_res + 1);
      W::for_n(P.n_res + 1, [&](int r) { rem[r] = lim[r]; });
    }
  }

  // Solve — scheduler.go:440-519 with Queue (queue.go:31-108)
  KS_FN void solve() {
    prefilter_templates();
    const int np = P.n_pods;
    const uint32_t cap = (uint32_t)np + 1;
    const uint32_t* sorted = P.sorted_pods;
    uint32_t* queue = S.queue;
    W::for_n(np, [&](int i) { queue[i] = sorted[i]; });
    uint32_t head = 0, tail = (uint32_t)np % cap, qlen = (uint32_t)np;
    long long steps = 0;
    int status = 0;
    while (qlen > 0) {
      int pod = (int)S.queue[head];
      if (S.last_len[pod] == qlen) break;                                   // queue.go:52-56
      if ((S.max_steps >= 0 && steps >= S.max_steps) || (S.cancel_flag && *S.cancel_flag)) { status = 2; break; }
      head = (head + 1) % cap; qlen--;
      steps++;
      ctr.queue_pops++;
      int rc = try_schedule(pod);
      if (rc < 0) { status = 1; break; }
      if (rc != E_OK) {
        W::store(&S.err[pod], (uint8_t)rc);
        W::store(&S.diag[pod], (uint8_t)last_diag);
        W::store(&S.queue[tail], (uint32_t)pod);
        tail = (tail + 1) % cap; qlen++;
        W::store(&S.last_len[pod], qlen);                                   // queue.go:63-66
        W::sync();
      } else {
        W::store(&S.err[pod], (uint8_t)0);
        W::store(&S.diag[pod], (uint8_t)0);
      }
    }
    ctr.slow_sorts = order.slow_sorts;
    W::store(S.n_claims_out, n_claims);
    if (status) W::store(S.status_out, status);
    if (W::leader()) *S.counters = ctr;
    W::sync();
  }
};

}  // namespace ks
