// nodecheck.h — ExistingNode.CanAdd (existingnode.go:81-106) for one (pod class, existing node) pair up to, not including,
// volume requirement alternatives and topology: taints, host ports, resources.Fits, strict Requirements.Compatible.
// One lane per node. Shared by the pack engine's existing-node scan (engine.h add_to_existing) and by ksolve_node_dead0,
// which evaluates every class against every PRISTINE node of a resident cluster once, for all the probes of a sweep.
#pragma once
#include "ksp.h"

namespace ks {

struct NodeClassCtx {   // the class side, wave-uniform
  const uint64_t* cls;       // hot class record (RecLayout k_*)
  const uint64_t* cls_cold;  // gte[nk] | lte[nk] | ...
  uint32_t kdef, kcomp, khg, khl;
  uint32_t kneg;             // keys on which the pod's operator is NotIn / DoesNotExist (may be undefined on the node, requirements.go:188)
  uint64_t ktol, khpc;       // taints the pod tolerates; host-port triples that match one of the pod's
  const int64_t* req;
};

// nz (has_nz): bit w = word w of the class's requirement masks is non-zero, when the caller has it at hand (the engine takes it with one
// ballot, a lane per word; walking the words here is a chain of dependent reads per key)
KS_FN NodeClassCtx node_class_ctx(const Dict& d, const RecLayout& ly, const uint64_t* cls, const uint64_t* cls_cold, uint64_t hp_conf, bool has_nz = false, uint64_t nz = 0) {
  NodeClassCtx x;
  x.cls = cls; x.cls_cold = cls_cold;
  x.kdef = (uint32_t)cls[ly.k_f0()]; x.kcomp = (uint32_t)(cls[ly.k_f0()] >> 32);
  x.khg = (uint32_t)cls[ly.k_f1()]; x.khl = (uint32_t)(cls[ly.k_f1()] >> 32);
  x.ktol = cls[ly.k_tol()]; x.khpc = hp_conf;
  x.req = (const int64_t*)(cls + ly.k_req());
  x.kneg = 0;
  for (uint32_t ks_ = x.kdef; ks_; ks_ &= ks_ - 1) {
    const int key = __builtin_ctz(ks_);
    bool ne_ = false;
    const uint32_t w0 = d.key_word_off[key], w1 = d.key_word_off[key + 1];
    if (has_nz) ne_ = w1 > w0 && ((nz >> w0) & (w1 - w0 >= 64 ? ~0ull : ((1ull << (w1 - w0)) - 1))) != 0;
    else for (uint32_t w = w0; w < w1; ++w) ne_ = ne_ || cls[ly.k_mask() + w] != 0;
    if (((x.kcomp >> key) & 1) ? ne_ : !ne_) x.kneg |= 1u << key;
  }
  return x;
}

// Where a node's mutable state is read from: the cluster's pristine tables (stride n_nodes, index = node) or a probe's
// overlay (stride ov_cap, index = overlay slot). Null hg: no bounds anywhere; null hp: no host ports in use.
struct NodeTabs {
  const uint64_t* mask;                      // [req_words][stride]
  const uint32_t *defined, *complement;      // [stride]
  const uint32_t *hg, *hl;                   // [stride] or null
  const int64_t *gte, *lte;                  // [n_keys][stride] or null
  const int64_t* remaining;                  // [n_res][stride]
  const uint64_t* hp;                        // [stride] or null
  size_t stride;
};

// A node's scalars as one lane reads them ONCE (ksolve_node_dead0 holds them in registers across its classes)
struct NodePre {
  uint64_t taints, hp;
  int64_t rem[kMaxRes];
  uint32_t ndef, ncomp, nhg, nhl;
};
KS_FN NodePre node_preload(const RecLayout& ly, uint64_t taints, const NodeTabs& t, size_t i) {
  NodePre n;
  n.taints = taints; n.hp = t.hp ? t.hp[i] : 0ull;
#pragma unroll
  for (int r = 0; r < kMaxRes; ++r) n.rem[r] = r < ly.nr ? t.remaining[(size_t)r * t.stride + i] : 0;
  n.ndef = t.defined[i]; n.ncomp = t.complement[i];
  n.nhg = t.hg ? t.hg[i] : 0u; n.nhl = t.hg ? t.hl[i] : 0u;             // bounds a Gt / Lt pod left on the node
  return n;
}
// the checks behind taints, host ports and resources.Fits: strict Requirements.Compatible of the pod's set with the node's
KS_FN bool node_reqs_ok(const Dict& d, const RecLayout& ly, const NodeClassCtx& x, uint32_t ndef, uint32_t ncomp, uint32_t nhg, uint32_t nhl, const NodeTabs& t, size_t i);
KS_FN bool node_static_ok(const Dict& d, const RecLayout& ly, const NodeClassCtx& x, uint64_t taints, const NodeTabs& t, size_t i) {
  if (taints & ~x.ktol) return false;                                              // taints — existingnode.go:83
  if (x.khpc && t.hp && (t.hp[i] & x.khpc)) return false;                          // host ports — existingnode.go:87-93
  bool fit = true;                                                                 // resources.Fits — :96
  for (int r = 0; r < ly.nr; ++r) { const int64_t rem = t.remaining[(size_t)r * t.stride + i]; fit = fit && rem >= 0 && x.req[r] <= rem; }
  if (!fit) return false;
  return node_reqs_ok(d, ly, x, t.defined[i], t.complement[i], t.hg ? t.hg[i] : 0u, t.hg ? t.hl[i] : 0u, t, i);   // (bounds a Gt / Lt pod left on the node)
}
KS_FN bool node_static_ok_pre(const Dict& d, const RecLayout& ly, const NodeClassCtx& x, const NodePre& n, const NodeTabs& t, size_t i) {
  if (n.taints & ~x.ktol) return false;
  if (x.khpc && (n.hp & x.khpc)) return false;
  bool fit = true;
#pragma unroll
  for (int r = 0; r < kMaxRes; ++r) if (r < ly.nr) { const int64_t rem = n.rem[r]; fit = fit && rem >= 0 && x.req[r] <= rem; }
  if (!fit) return false;
  return node_reqs_ok(d, ly, x, n.ndef, n.ncomp, n.nhg, n.nhl, t, i);
}
KS_FN bool node_reqs_ok(const Dict& d, const RecLayout& ly, const NodeClassCtx& x, uint32_t ndef, uint32_t ncomp, uint32_t nhg, uint32_t nhl, const NodeTabs& t, size_t i) {
  if (x.kdef & ~ndef & ~x.kneg) return false;                                      // undefined key — requirements.go:185-193
  for (uint32_t both = x.kdef & ndef; both; both &= both - 1) {                    // Intersects — requirements.go:254-274
    const int key = __builtin_ctz(both);
    const bool ca = (ncomp >> key) & 1, cb = (x.kcomp >> key) & 1;
    bool hg = (x.khg >> key) & 1, hl = (x.khl >> key) & 1;
    int64_t g = hg ? ((const int64_t*)x.cls_cold)[key] : 0, lq = hl ? ((const int64_t*)x.cls_cold)[ly.nk + key] : 0;
    if ((nhg >> key) & 1) { const int64_t v = t.gte[(size_t)key * t.stride + i]; g = hg && g > v ? g : v; hg = true; }   // maxIntPtr / minIntPtr — requirement.go:352-376
    if ((nhl >> key) & 1) { const int64_t v = t.lte[(size_t)key * t.stride + i]; lq = hl && lq < v ? lq : v; hl = true; }
    const bool empty_bounds = hg && hl && g > lq;                                  // HasIntersection — requirement.go:220-224
    if (ca && cb && !empty_bounds) continue;
    bool hit = false, nonempty_n = false;
    for (uint32_t w = d.key_word_off[key]; w < d.key_word_off[key + 1]; ++w) {
      const uint64_t a = t.mask[(size_t)w * t.stride + i], b = x.cls[ly.k_mask() + w];
      nonempty_n = nonempty_n || a != 0;
      uint64_t c = ca ? (b & ~a) : cb ? (a & ~b) : (a & b);
      if (c && (hg || hl)) c = inbounds_word(d, w, c, hg, g, hl, lq);
      hit = hit || c != 0;
    }
    if (hit && !empty_bounds) continue;
    const bool neg_n = ca ? nonempty_n : !nonempty_n;
    if (neg_n && ((x.kneg >> key) & 1)) continue;
    return false;
  }
  return true;
}

}  // namespace ks
