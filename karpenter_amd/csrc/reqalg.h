// reqalg.h — requirement-set algebra on the flat (dictionary-encoded bitmask) representation.
//
// Semantics follow pkg/scheduling/requirement.go and requirements.go of the reference (cited per function). A
// requirement on key k is (complement, value bitmask over k's dictionary, optional inclusive int bounds, optional
// minValues); every value that can appear anywhere in a problem is in the dictionary, so concrete sets are exact
// and a complement set means "everything except these dictionary values" (including values outside the dictionary,
// which is why two complements always intersect — requirement.go:226-228).
#pragma once
#include "wave.h"

namespace ks {

constexpr int kMaxKeys = 32;
constexpr int kMaxReqWords = 96;   // sum over keys of dictionary words (instance-type key may take up to 32)
constexpr int kMaxRes = 8;
constexpr int kMaxItWords = 32;

struct Dict {
  int n_keys;
  int req_words;
  uint32_t key_word_off[kMaxKeys + 1];
  uint32_t well_known_mask;
  int key_it, key_zone, key_ct, key_hostname;
  const int64_t* value_int;     // req_words*64
  const uint64_t* value_is_int; // req_words
  const uint64_t* value_valid;  // req_words : bits that are real dictionary entries
};

// One requirement set living in SoA tables (pods, templates, instance types, nodes, claims).
struct ReqRef {
  const uint64_t* mask;  // req_words
  uint32_t defined, complement, has_gte, has_lte;
  const int64_t* gte;    // n_keys or nullptr
  const int64_t* lte;
  const int32_t* minv;   // n_keys or nullptr (-1 = nil)
};

// Working copy (the "nodeClaimRequirements" being built in NodeClaim.CanAdd, nodeclaim.go:130-208).
// A requirement set being built (mask words + bounds + minValues). RW = mask words it holds: kMaxReqWords for any problem the
// library accepts; the consolidation sweep's compact scratch (engine.h ScratchT) uses a smaller bound when the problem's
// dictionaries allow it — the code only ever touches the first Dict::req_words words.
template <int RW>
struct ReqBufT {
  uint64_t mask[RW];
  uint32_t defined, complement, has_gte, has_lte, has_minv;
  int64_t gte[kMaxKeys], lte[kMaxKeys];
  int32_t minv[kMaxKeys];

  KS_FN ReqRef ref() const {
    ReqRef r;
    r.mask = mask; r.defined = defined; r.complement = complement; r.has_gte = has_gte; r.has_lte = has_lte;
    r.gte = gte; r.lte = lte; r.minv = has_minv ? minv : nullptr;
    return r;
  }
};
typedef ReqBufT<kMaxReqWords> ReqBuf;

KS_FN bool bit(uint32_t m, int k) { return (m >> k) & 1u; }

// Operator() — requirement.go:290-301. 0 In, 1 NotIn, 2 Exists, 3 DoesNotExist
enum { OP_IN = 0, OP_NOTIN = 1, OP_EXISTS = 2, OP_DNE = 3 };
KS_FN bool key_nonempty(const Dict& d, const uint64_t* mask, int k) {
  for (uint32_t w = d.key_word_off[k]; w < d.key_word_off[k + 1]; ++w) if (mask[w]) return true;
  return false;
}
KS_FN int req_op(const Dict& d, const ReqRef& r, int k) {
  bool ne = key_nonempty(d, r.mask, k);
  if (bit(r.complement, k)) return ne ? OP_NOTIN : OP_EXISTS;
  return ne ? OP_IN : OP_DNE;
}
KS_FN bool op_negative(int op) { return op == OP_NOTIN || op == OP_DNE; }

// withinBounds — requirement.go:334-350, as a mask over one dictionary word
KS_FN uint64_t inbounds_word(const Dict& d, uint32_t w, uint64_t candidates, bool hg, int64_t g, bool hl, int64_t l) {
  if (!hg && !hl) return candidates;
  uint64_t c = candidates & d.value_is_int[w];  // non-integers are out of bounds once any bound is set
  uint64_t out = 0;
  while (c) {
    int b = ctz64(c);
    c &= c - 1;
    int64_t v = d.value_int[(size_t)w * 64 + b];
    if ((!hg || v >= g) && (!hl || v <= l)) out |= 1ull << b;
  }
  return out;
}

struct Bounds { bool hg, hl; int64_t g, l; };
// maxIntPtr / minIntPtr — requirement.go:352-376
KS_FN Bounds combine_bounds(const ReqRef& a, const ReqRef& b, int k) {
  Bounds o;
  bool ag = bit(a.has_gte, k), bg = bit(b.has_gte, k), al = bit(a.has_lte, k), bl = bit(b.has_lte, k);
  int64_t agv = ag ? a.gte[k] : 0, bgv = bg ? b.gte[k] : 0, alv = al ? a.lte[k] : 0, blv = bl ? b.lte[k] : 0;
  o.hg = ag || bg; o.g = ag && bg ? (agv > bgv ? agv : bgv) : (ag ? agv : bgv);
  o.hl = al || bl; o.l = al && bl ? (alv < blv ? alv : blv) : (al ? alv : blv);
  return o;
}

// HasIntersection — requirement.go:220-254 (receiver a, argument b; symmetric)
KS_FN bool has_intersection(const Dict& d, const ReqRef& a, const ReqRef& b, int k) {
  Bounds bd = combine_bounds(a, b, k);
  if (bd.hg && bd.hl && bd.g > bd.l) return false;
  bool ac = bit(a.complement, k), bc = bit(b.complement, k);
  if (ac && bc) return true;
  for (uint32_t w = d.key_word_off[k]; w < d.key_word_off[k + 1]; ++w) {
    uint64_t c = ac ? (b.mask[w] & ~a.mask[w]) : bc ? (a.mask[w] & ~b.mask[w]) : (a.mask[w] & b.mask[w]);
    if (c && inbounds_word(d, w, c, bd.hg, bd.g, bd.hl, bd.l)) return true;
  }
  return false;
}

// Has(value) — requirement.go:275-280, for dictionary value (word w, bit b) of key k
KS_FN bool req_has(const Dict& d, const ReqRef& r, int k, uint32_t w, int b) {
  bool in = (r.mask[w] >> b) & 1ull;
  bool hg = bit(r.has_gte, k), hl = bit(r.has_lte, k);
  if (hg || hl) {
    if (!((d.value_is_int[w] >> b) & 1ull)) return false;
    int64_t v = d.value_int[(size_t)w * 64 + b];
    if (hg && v < r.gte[k]) return false;
    if (hl && v > r.lte[k]) return false;
  }
  return bit(r.complement, k) ? !in : in;
}

enum { COMPAT_OK = 0, COMPAT_UNDEFINED_KEY = 1, COMPAT_NO_INTERSECTION = 2 };

// Intersects — requirements.go:254-274 (existing = r, incoming = q)
KS_FN bool reqs_intersect(const Dict& d, const ReqRef& r, const ReqRef& q) {
  uint32_t both = r.defined & q.defined;
  while (both) {
    int k = __builtin_ctz(both);
    both &= both - 1;
    if (!has_intersection(d, r, q, k)) {
      if (op_negative(req_op(d, q, k)) && op_negative(req_op(d, r, k))) continue;
      return false;
    }
  }
  return true;
}
// Compatible — requirements.go:181-197. allow_undefined = AllowUndefinedWellKnownLabels
KS_FN int reqs_compatible(const Dict& d, const ReqRef& r, const ReqRef& q, bool allow_undefined) {
  uint32_t undef = q.defined & ~r.defined;
  if (allow_undefined) undef &= ~d.well_known_mask;
  while (undef) {
    int k = __builtin_ctz(undef);
    undef &= undef - 1;
    if (!op_negative(req_op(d, q, k))) return COMPAT_UNDEFINED_KEY;
  }
  return reqs_intersect(d, r, q) ? COMPAT_OK : COMPAT_NO_INTERSECTION;
}

template <class RB>
KS_FN void reqbuf_load(const Dict& d, RB& out, const ReqRef& r) {
  for (int w = 0; w < d.req_words; ++w) out.mask[w] = r.mask[w];
  out.defined = r.defined; out.complement = r.complement; out.has_gte = r.has_gte; out.has_lte = r.has_lte;
  out.has_minv = 0;
  for (int k = 0; k < d.n_keys; ++k) {
    out.gte[k] = (r.gte && bit(r.has_gte, k)) ? r.gte[k] : 0;
    out.lte[k] = (r.lte && bit(r.has_lte, k)) ? r.lte[k] : 0;
    int32_t mv = r.minv ? r.minv[k] : -1;
    out.minv[k] = mv;
    if (mv >= 0) out.has_minv |= 1u << k;
  }
}

// Requirements.Add(q...) — requirements.go:133-140 with Requirement.Intersection — requirement.go:181-214.
// Returns true when any key of `acc` changed (used to know when a claim's requirement state moved).
template <class RB>
KS_FN bool reqbuf_add(const Dict& d, RB& acc, const ReqRef& q) {
  bool changed = false;
  uint32_t keys = q.defined;
  while (keys) {
    int k = __builtin_ctz(keys);
    keys &= keys - 1;
    uint32_t kb = 1u << k;
    uint32_t w0 = d.key_word_off[k], w1 = d.key_word_off[k + 1];
    int32_t qmv = q.minv ? q.minv[k] : -1;
    if (!(acc.defined & kb)) {
      for (uint32_t w = w0; w < w1; ++w) acc.mask[w] = q.mask[w];
      acc.defined |= kb;
      acc.complement = (acc.complement & ~kb) | (q.complement & kb);
      acc.has_gte = (acc.has_gte & ~kb) | (q.has_gte & kb);
      acc.has_lte = (acc.has_lte & ~kb) | (q.has_lte & kb);
      acc.gte[k] = bit(q.has_gte, k) ? q.gte[k] : 0;
      acc.lte[k] = bit(q.has_lte, k) ? q.lte[k] : 0;
      acc.minv[k] = qmv;
      if (qmv >= 0) acc.has_minv |= kb;
      changed = true;
      continue;
    }
    ReqRef a = acc.ref();
    a.minv = acc.minv;
    Bounds bd = combine_bounds(a, q, k);
    bool ac = bit(acc.complement, k), qc = bit(q.complement, k);
    bool comp = ac && qc;
    int32_t mv = acc.minv[k] > qmv ? acc.minv[k] : qmv;
    bool old_comp = ac, old_hg = bit(acc.has_gte, k), old_hl = bit(acc.has_lte, k);
    int64_t old_g = acc.gte[k], old_l = acc.lte[k];
    int32_t old_mv = acc.minv[k];
    bool masks_changed = false;
    if (bd.hg && bd.hl && bd.g > bd.l) {
      // NewRequirementWithFlexibility(key, DoesNotExist, minValues) — requirement.go:189-191
      for (uint32_t w = w0; w < w1; ++w) { if (acc.mask[w]) masks_changed = true; acc.mask[w] = 0; }
      comp = false; bd.hg = bd.hl = false;
    } else {
      for (uint32_t w = w0; w < w1; ++w) {
        uint64_t am = acc.mask[w], qm = q.mask[w];
        uint64_t v = (ac && qc) ? (am | qm) : (ac && !qc) ? (qm & ~am) : (!ac && qc) ? (am & ~qm) : (am & qm);
        v = inbounds_word(d, w, v, bd.hg, bd.g, bd.hl, bd.l);
        if (v != am) masks_changed = true;
        acc.mask[w] = v;
      }
      if (!comp) bd.hg = bd.hl = false;  // remove boundaries for concrete sets — requirement.go:209-212
    }
    acc.complement = comp ? (acc.complement | kb) : (acc.complement & ~kb);
    acc.has_gte = bd.hg ? (acc.has_gte | kb) : (acc.has_gte & ~kb);
    acc.has_lte = bd.hl ? (acc.has_lte | kb) : (acc.has_lte & ~kb);
    acc.gte[k] = bd.hg ? bd.g : 0;
    acc.lte[k] = bd.hl ? bd.l : 0;
    acc.minv[k] = mv;
    if (mv >= 0) acc.has_minv |= kb;
    if (masks_changed || comp != old_comp || bd.hg != old_hg || bd.hl != old_hl || (bd.hg && bd.g != old_g) || (bd.hl && bd.l != old_l) || mv != old_mv)
      changed = true;
  }
  return changed;
}

}  // namespace ks
