// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// CPU restatement of the scheduling algebra of kubernetes-sigs/karpenter (reference @ /root/reference):
//   pkg/scheduling/requirement.go, requirements.go, taints.go ; pkg/utils/resources/resources.go
// plus the third-party behaviour the reference calls into and that is NOT under /root/reference
// (restated from the published semantics):
//   k8s.io/apimachinery v0.36.1 resource.Quantity (exact decimal)    -> Quantity below (int128 nano-units, exact)
//   k8s.io/api v0.36.1 core/v1 Toleration.ToleratesTaint(.., true)   -> tolerates_taint below   [parity unpinned at unit level]
//   k8s.io/apimachinery labels.Selector                                -> LabelSelector::matches
// Every function cites the reference file:line it follows. Value sets are real sets of (interned) strings on purpose:
// this file must stay an independent restatement of the reference, not a copy of the product's bitmask encoding
// (intern.hpp says what interning changes — the containers — and what it does not — the algorithm).
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <map>
#include <optional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "intern.hpp"

namespace oracle {

typedef __int128 i128;

// The strings the path names literally, interned once.
struct WellKnownSyms {
  Sym hostname, zone, region, instance_type, arch, os, windows_build, nodepool, capacity_type, initialized, registered, reservation_id;
  Sym pods, cpu, memory, nodes;
  Sym empty, Equal, Exists, Lt, Gt;
  Sym PreferNoSchedule, Honor, Ignore, DoNotSchedule, ScheduleAnyway, Pending, Failed, Succeeded, reserved, spot, on_demand, true_;
  Sym ip_any4, ip_any6;
};
inline const WellKnownSyms& W();

// ---------------------------------------------------------------------------------------------------------------
// resource.Quantity (k8s.io/apimachinery/pkg/api/resource) — exact decimal. Stored as int128 nano-units, which is
// exact for every quantity Kubernetes can represent (its own finest scale is nano).
// ---------------------------------------------------------------------------------------------------------------
inline i128 parse_quantity(const std::string& s) {
  if (s.empty()) throw std::runtime_error("quantity: empty");
  size_t i = 0;
  bool neg = false;
  if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; ++i; }
  i128 mant = 0;
  int frac_digits = 0;
  bool seen_dot = false, any = false;
  for (; i < s.size(); ++i) {
    char c = s[i];
    if (c >= '0' && c <= '9') { mant = mant * 10 + (c - '0'); if (seen_dot) ++frac_digits; any = true; }
    else if (c == '.' && !seen_dot) seen_dot = true;
    else break;
  }
  if (!any) throw std::runtime_error("quantity: no digits in '" + s + "'");
  std::string suf = s.substr(i);
  // value = mant * 10^-frac_digits * mult ; result in nano-units (10^-9)
  i128 bin = 1;
  int dec = 0;
  if (suf == "") dec = 0;
  else if (suf == "n") dec = -9; else if (suf == "u") dec = -6; else if (suf == "m") dec = -3;
  else if (suf == "k") dec = 3; else if (suf == "M") dec = 6; else if (suf == "G") dec = 9;
  else if (suf == "T") dec = 12; else if (suf == "P") dec = 15; else if (suf == "E") dec = 18;
  else if (suf == "Ki") bin = (i128)1 << 10; else if (suf == "Mi") bin = (i128)1 << 20; else if (suf == "Gi") bin = (i128)1 << 30;
  else if (suf == "Ti") bin = (i128)1 << 40; else if (suf == "Pi") bin = (i128)1 << 50; else if (suf == "Ei") bin = (i128)1 << 60;
  else if (suf[0] == 'e' || suf[0] == 'E') dec = atoi(suf.c_str() + 1);
  else throw std::runtime_error("quantity: bad suffix '" + suf + "'");
  int e = dec - frac_digits + 9;  // exponent to reach nano-units
  i128 v = mant * bin;
  for (; e > 0; --e) v *= 10;
  for (; e < 0; ++e) {
    // Kubernetes rounds up anything finer than nano; the oracle rejects it instead (never silently inexact).
    if (v % 10 != 0) throw std::runtime_error("quantity: finer than nano '" + s + "'");
    v /= 10;
  }
  return neg ? -v : v;
}

// v1.ResourceList: resource name -> quantity, a small array sorted by the name's symbol
struct ResEntry { Sym first; i128 second; };
class ResourceList {
  SmallVec<ResEntry, 5> v_;
  size_t lower(Sym k) const { size_t i = 0; while (i < v_.size() && v_[i].first < k) ++i; return i; }

 public:
  typedef ResEntry* iterator;
  typedef const ResEntry* const_iterator;
  iterator begin() { return v_.begin(); }
  iterator end() { return v_.end(); }
  const_iterator begin() const { return v_.begin(); }
  const_iterator end() const { return v_.end(); }
  size_t size() const { return v_.size(); }
  bool empty() const { return v_.empty(); }
  const_iterator find(Sym k) const { size_t i = lower(k); return i < v_.size() && v_[i].first == k ? v_.begin() + i : v_.end(); }
  iterator find(Sym k) { size_t i = lower(k); return i < v_.size() && v_[i].first == k ? v_.begin() + i : v_.end(); }
  bool count(Sym k) const { return find(k) != end(); }
  i128& operator[](Sym k) {
    size_t i = lower(k);
    if (i == v_.size() || v_[i].first != k) v_.insert_at(i, ResEntry{k, 0});
    return v_[i].second;
  }
  iterator erase(iterator it) { size_t i = it - v_.begin(); v_.erase_at(i); return v_.begin() + i; }
  bool operator==(const ResourceList& o) const {
    if (v_.size() != o.v_.size()) return false;
    for (size_t i = 0; i < v_.size(); ++i) if (v_[i].first != o.v_[i].first || v_[i].second != o.v_[i].second) return false;
    return true;
  }
};

// resources.Merge — pkg/utils/resources/resources.go:52-66
inline ResourceList res_merge(const ResourceList& a, const ResourceList& b) {
  ResourceList out = a;
  for (auto& kv : b) out[kv.first] += kv.second;
  return out;
}
// resources.Subtract — resources.go:83-97 (keys of lhs only)
inline ResourceList res_subtract(const ResourceList& lhs, const ResourceList& rhs) {
  ResourceList out;
  for (auto& kv : lhs) {
    i128 v = kv.second;
    auto it = rhs.find(kv.first);
    if (it != rhs.end()) v -= it->second;
    out[kv.first] = v;
  }
  return out;
}
// resources.SubtractFrom — resources.go:99-110 (keys of src, in place)
inline void res_subtract_from(ResourceList& dest, const ResourceList& src) {
  for (auto& kv : src) dest[kv.first] -= kv.second;
}
// resources.MaxResources — resources.go:130-141
inline ResourceList res_max(const std::vector<const ResourceList*>& lists) {
  ResourceList out;
  for (auto* l : lists)
    for (auto& kv : *l) {
      auto it = out.find(kv.first);
      if (it == out.end() || kv.second > it->second) out[kv.first] = kv.second;
    }
  return out;
}
// resources.MinResources — resources.go:145-170 (intersection of keys)
inline ResourceList res_min(const std::vector<const ResourceList*>& lists) {
  ResourceList out;
  if (lists.empty()) return out;
  out = *lists[0];
  for (size_t i = 1; i < lists.size(); ++i) {
    for (auto it = out.begin(); it != out.end();) {
      auto jt = lists[i]->find(it->first);
      if (jt == lists[i]->end()) it = out.erase(it);
      else { if (jt->second < it->second) it->second = jt->second; ++it; }
    }
  }
  return out;
}
// resources.Fits — resources.go:188-201
inline bool res_fits(const ResourceList& candidate, const ResourceList& total) {
  for (auto& kv : total) if (kv.second < 0) return false;
  for (auto& kv : candidate) {
    auto it = total.find(kv.first);
    i128 t = it == total.end() ? 0 : it->second;
    if (kv.second > t) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Well-known labels — pkg/apis/v1/labels.go:40-134. Providers extend WellKnownLabels at init
// (kwok/apis/v1alpha1/labels.go:40, pkg/cloudprovider/fake/instancetype.go:46); the problem file lists the extras.
// ---------------------------------------------------------------------------------------------------------------
static const char* kLabelHostname = "kubernetes.io/hostname";
static const char* kLabelZone = "topology.kubernetes.io/zone";
static const char* kLabelRegion = "topology.kubernetes.io/region";
static const char* kLabelInstanceType = "node.kubernetes.io/instance-type";
static const char* kLabelArch = "kubernetes.io/arch";
static const char* kLabelOS = "kubernetes.io/os";
static const char* kLabelWindowsBuild = "node.kubernetes.io/windows-build";
static const char* kNodePoolLabel = "karpenter.sh/nodepool";
static const char* kCapacityTypeLabel = "karpenter.sh/capacity-type";
static const char* kNodeInitializedLabel = "karpenter.sh/initialized";
static const char* kNodeRegisteredLabel = "karpenter.sh/registered";
static const char* kReservationIDLabel = "karpenter.sh/reservation-id";  // cloudprovider.ReservationIDLabel
static const char* kMinValuesRelaxedAnnotation = "karpenter.sh/nodeclaim-min-values-relaxed";

inline const WellKnownSyms& W() {
  static const WellKnownSyms w = [] {
    WellKnownSyms x;
    x.hostname = sym(kLabelHostname); x.zone = sym(kLabelZone); x.region = sym(kLabelRegion); x.instance_type = sym(kLabelInstanceType);
    x.arch = sym(kLabelArch); x.os = sym(kLabelOS); x.windows_build = sym(kLabelWindowsBuild); x.nodepool = sym(kNodePoolLabel);
    x.capacity_type = sym(kCapacityTypeLabel); x.initialized = sym(kNodeInitializedLabel); x.registered = sym(kNodeRegisteredLabel);
    x.reservation_id = sym(kReservationIDLabel);
    x.pods = sym("pods"); x.cpu = sym("cpu"); x.memory = sym("memory"); x.nodes = sym("nodes");
    x.empty = sym(""); x.Equal = sym("Equal"); x.Exists = sym("Exists"); x.Lt = sym("Lt"); x.Gt = sym("Gt");
    x.PreferNoSchedule = sym("PreferNoSchedule"); x.Honor = sym("Honor"); x.Ignore = sym("Ignore");
    x.DoNotSchedule = sym("DoNotSchedule"); x.ScheduleAnyway = sym("ScheduleAnyway");
    x.Pending = sym("Pending"); x.Failed = sym("Failed"); x.Succeeded = sym("Succeeded");
    x.reserved = sym("reserved"); x.spot = sym("spot"); x.on_demand = sym("on-demand"); x.true_ = sym("true");
    x.ip_any4 = sym("0.0.0.0"); x.ip_any6 = sym("::");
    return x;
  }();
  return w;
}

struct Labels {
  SymSet well_known{W().nodepool, W().zone, W().region, W().instance_type, W().arch, W().os, W().capacity_type, W().windows_build};
  // NormalizedLabels — labels.go:121-127
  SymMap normalized;
  Labels() {
    normalized.set(sym("failure-domain.beta.kubernetes.io/zone"), W().zone);
    normalized.set(sym("beta.kubernetes.io/arch"), W().arch);
    normalized.set(sym("beta.kubernetes.io/os"), W().os);
    normalized.set(sym("beta.kubernetes.io/instance-type"), W().instance_type);
    normalized.set(sym("failure-domain.beta.kubernetes.io/region"), W().region);
  }
};
inline Labels& labels_registry() { static Labels l; return l; }

// strconv.Atoi as used by withinBounds (requirement.go:339): optional sign, decimal digits, int64 range.
inline bool go_atoi(const std::string& s, long long& out) {
  if (s.empty()) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; }
  if (i == s.size()) return false;
  unsigned long long v = 0;
  for (; i < s.size(); ++i) {
    char c = s[i];
    if (c == '_' ) return false;
    if (c < '0' || c > '9') return false;
    unsigned d = c - '0';
    if (v > (ULLONG_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  if (!neg && v > (unsigned long long)LLONG_MAX) return false;
  if (neg && v > (unsigned long long)LLONG_MAX + 1ULL) return false;
  out = neg ? (long long)(0 - v) : (long long)v;
  return true;
}

enum class Op { In, NotIn, Exists, DoesNotExist, Gt, Lt, Gte, Lte };
inline Op parse_op(const std::string& s) {
  if (s == "In") return Op::In;
  if (s == "NotIn") return Op::NotIn;
  if (s == "Exists") return Op::Exists;
  if (s == "DoesNotExist") return Op::DoesNotExist;
  if (s == "Gt") return Op::Gt;
  if (s == "Lt") return Op::Lt;
  if (s == "Gte") return Op::Gte;
  if (s == "Lte") return Op::Lte;
  throw std::runtime_error("bad operator " + s);
}
inline const char* op_name(Op o) {
  switch (o) {
    case Op::In: return "In"; case Op::NotIn: return "NotIn"; case Op::Exists: return "Exists";
    case Op::DoesNotExist: return "DoesNotExist"; case Op::Gt: return "Gt"; case Op::Lt: return "Lt";
    case Op::Gte: return "Gte"; case Op::Lte: return "Lte";
  }
  return "";
}

typedef std::optional<long long> OptInt;

// withinBounds — requirement.go:334-350
inline bool within_bounds(Sym v, const OptInt& gte, const OptInt& lte) {
  if (!gte && !lte) return true;
  long long val;
  if (!go_atoi(str(v), val)) return false;
  if (gte && val < *gte) return false;
  if (lte && val > *lte) return false;
  return true;
}
inline OptInt min_opt(const OptInt& a, const OptInt& b) { if (!a) return b; if (!b) return a; return *a < *b ? a : b; }  // requirement.go:352-363
inline OptInt max_opt(const OptInt& a, const OptInt& b) { if (!a) return b; if (!b) return a; return *a > *b ? a : b; }  // requirement.go:365-376

// Requirement — requirement.go:36-43
struct Requirement {
  Sym key = kNoSym;
  bool complement = false;
  SymSet values;
  OptInt gte, lte;
  std::optional<int> min_values;

  // NewRequirementWithFlexibility — requirement.go:48-110
  static Requirement make(Sym key, Op op, std::optional<int> min_values, const Sym* vals, size_t n_vals) {
    if (auto* nz = labels_registry().normalized.find(key)) key = nz->second;
    Requirement r;
    r.key = key;
    r.min_values = min_values;
    if (op == Op::In) { r.values.insert(vals, vals + n_vals); r.complement = false; return r; }
    r.complement = true;
    if (op == Op::DoesNotExist) r.complement = false;
    if (op == Op::NotIn) r.values.insert(vals, vals + n_vals);
    auto atoi0 = [&](Sym s) { long long v = 0; go_atoi(str(s), v); return v; };
    if (op == Op::Gt) {
      long long v = n_vals == 0 ? 0 : atoi0(vals[0]);
      if (v == LLONG_MAX) return make(key, Op::DoesNotExist, std::nullopt, nullptr, 0);  // requirement.go:85-88 (minValues dropped)
      r.gte = v + 1;
    }
    if (op == Op::Lt) { long long v = n_vals == 0 ? 0 : atoi0(vals[0]); r.lte = v - 1; }
    if (op == Op::Gte) r.gte = n_vals == 0 ? 0 : atoi0(vals[0]);
    if (op == Op::Lte) r.lte = n_vals == 0 ? 0 : atoi0(vals[0]);
    return r;
  }
  static Requirement make(Sym key, Op op) { return make(key, op, std::nullopt, nullptr, 0); }
  static Requirement make(Sym key, Op op, Sym one) { return make(key, op, std::nullopt, &one, 1); }
  static Requirement make(Sym key, Op op, std::optional<int> min_values, const std::vector<Sym>& vals) { return make(key, op, min_values, vals.data(), vals.size()); }
  // from strings (parsing, tests)
  static Requirement make_s(const std::string& key, Op op, std::optional<int> min_values, const std::vector<std::string>& vals) {
    std::vector<Sym> v;
    for (auto& x : vals) v.push_back(sym(x));
    return make(sym(key), op, min_values, v);
  }

  // Len — requirement.go:303-308 (MaxInt64 - |values| for complements)
  long long len() const { return complement ? LLONG_MAX - (long long)values.size() : (long long)values.size(); }
  // Operator — requirement.go:290-301
  Op op() const {
    if (complement) return len() < LLONG_MAX ? Op::NotIn : Op::Exists;
    return len() > 0 ? Op::In : Op::DoesNotExist;
  }
  // Has — requirement.go:275-280
  bool has(Sym v) const {
    if (complement) return !values.count(v) && within_bounds(v, gte, lte);
    return values.count(v) && within_bounds(v, gte, lte);
  }
  // Intersection — requirement.go:181-214  (receiver r, argument q)
  Requirement intersection(const Requirement& q) const {
    bool comp = complement && q.complement;
    OptInt g = max_opt(gte, q.gte), l = min_opt(lte, q.lte);
    std::optional<int> mv;
    if (min_values && q.min_values) mv = std::max(*min_values, *q.min_values);
    else mv = min_values ? min_values : q.min_values;
    if (g && l && *g > *l) return make(key, Op::DoesNotExist, mv, nullptr, 0);
    Requirement out;
    SymSet& vals = out.values;
    const bool bounded = g || l;
    if (complement && q.complement) { vals = values; vals.insert(q.values.begin(), q.values.end()); }
    else if (complement && !q.complement) { for (Sym v : q.values) if (!values.count(v)) vals.append_sorted(v); }
    else if (!complement && q.complement) { for (Sym v : values) if (!q.values.count(v)) vals.append_sorted(v); }
    else { for (Sym v : values) if (q.values.count(v)) vals.append_sorted(v); }
    if (bounded) {
      SymSet kept;
      for (Sym v : vals) if (within_bounds(v, g, l)) kept.append_sorted(v);
      vals = kept;
    }
    if (!comp) { g.reset(); l.reset(); }
    out.key = key; out.complement = comp; out.gte = g; out.lte = l; out.min_values = mv;
    return out;
  }
  // HasIntersection — requirement.go:220-254
  bool has_intersection(const Requirement& q) const {
    OptInt g = max_opt(gte, q.gte), l = min_opt(lte, q.lte);
    if (g && l && *g > *l) return false;
    if (complement && q.complement) return true;
    if (complement && !q.complement) { for (Sym v : q.values) if (!values.count(v) && within_bounds(v, g, l)) return true; return false; }
    if (!complement && q.complement) { for (Sym v : values) if (!q.values.count(v) && within_bounds(v, g, l)) return true; return false; }
    for (Sym v : values) if (q.values.count(v) && within_bounds(v, g, l)) return true;
    return false;
  }
  // Any — requirement.go:256-271. The reference draws a random element; callers in scope only use it on
  // single-valued requirements (Offering.Zone()/CapacityType()). Canonicalised: lexicographically smallest value.
  Sym any() const {
    if (op() == Op::In) return values.lex_min();
    return W().empty;
  }
  bool operator==(const Requirement& o) const {
    return key == o.key && complement == o.complement && values == o.values && gte == o.gte && lte == o.lte && min_values == o.min_values;
  }
};

// Requirements — requirements.go:34-47: key -> Requirement, kept as an array sorted by the key's symbol. No result depends on
// the order the reference's map is walked in (error texts aside, which are Go-map-order dependent there too).
struct Requirements {
  std::vector<Requirement> m;
  SmallVec<Sym, 12> keys;   // m[i].key, side by side: looking a key up does not walk the 88-byte records

  size_t lower(Sym k) const { size_t i = 0; const size_t n = keys.size(); while (i < n && keys[i] < k) ++i; return i; }
  const Requirement* find(Sym k) const { size_t i = lower(k); return i < keys.size() && keys[i] == k ? &m[i] : nullptr; }
  Requirement* find(Sym k) { size_t i = lower(k); return i < keys.size() && keys[i] == k ? &m[i] : nullptr; }
  void put(const Requirement& r) {   // map assignment
    size_t i = lower(r.key);
    if (i < keys.size() && keys[i] == r.key) m[i] = r; else { m.insert(m.begin() + i, r); keys.insert_at(i, r.key); }
  }
  void erase(Sym k) { size_t i = lower(k); if (i < keys.size() && keys[i] == k) { m.erase(m.begin() + i); keys.erase_at(i); } }
  // Add — requirements.go:133-140 : incoming.Intersection(existing)
  void add(const Requirement& in) {
    size_t i = lower(in.key);
    if (i < keys.size() && keys[i] == in.key) m[i] = in.intersection(m[i]);
    else { m.insert(m.begin() + i, in); keys.insert_at(i, in.key); }
  }
  void add_all(const Requirements& o) { for (auto& r : o.m) add(r); }
  bool has(Sym k) const { return find(k) != nullptr; }
  // Get — requirements.go:160-166 (undefined => Exists)
  Requirement get(Sym k) const {
    const Requirement* r = find(k);
    if (!r) return Requirement::make(k, Op::Exists);
    return *r;
  }
  bool has_min_values() const { for (auto& r : m) if (r.min_values) return true; return false; }  // requirements.go:276-283

  // Intersects — requirements.go:254-274 ; on failure *bad_key = an offending key
  bool intersects(const Requirements& in, Sym* bad_key = nullptr) const {
    size_t j = 0;
    const size_t ni = in.keys.size();
    for (size_t i = 0; i < keys.size(); ++i) {
      const Sym key = keys[i];
      while (j < ni && in.keys[j] < key) ++j;
      if (j == ni) break;
      if (in.keys[j] != key) continue;
      const Requirement& existing = m[i];
      const Requirement& incoming = in.m[j];
      if (!existing.has_intersection(incoming)) {
        Op oi = incoming.op();
        if (oi == Op::NotIn || oi == Op::DoesNotExist) {
          Op oe = existing.op();
          if (oe == Op::NotIn || oe == Op::DoesNotExist) continue;
        }
        if (bad_key) *bad_key = key;
        return false;
      }
    }
    return true;
  }
  // Compatible — requirements.go:181-197 ; allow_undefined = AllowUndefinedWellKnownLabels when true
  bool compatible(const Requirements& in, bool allow_undefined_well_known, std::string* why = nullptr) const {
    auto& wk = labels_registry().well_known;
    for (auto& r : in.m) {
      if (allow_undefined_well_known && wk.count(r.key)) continue;
      if (has(r.key)) continue;
      Op o = r.op();
      if (o == Op::NotIn || o == Op::DoesNotExist) continue;
      if (why) *why = "label \"" + str(r.key) + "\" does not have known values";
      return false;
    }
    Sym bad = kNoSym;
    if (!intersects(in, &bad)) { if (why) *why = "key " + str(bad) + " incompatible"; return false; }
    return true;
  }
};

// NewLabelRequirements — requirements.go:67-73
inline Requirements label_requirements(const SymMap& labels) {
  Requirements r;
  for (auto& kv : labels) r.add(Requirement::make(kv.first, Op::In, kv.second));
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// Taints — pkg/scheduling/taints.go:78-95 and k8s.io/api core/v1 Toleration.ToleratesTaint
// ---------------------------------------------------------------------------------------------------------------
struct Taint { Sym key, value, effect; };
struct Toleration { Sym key, op, value, effect; };

inline bool tolerates_taint(const Toleration& t, const Taint& taint) {
  const WellKnownSyms& w = W();
  if (t.effect != w.empty && t.effect != taint.effect) return false;
  if (t.key != w.empty && t.key != taint.key) return false;
  if (t.op == w.empty || t.op == w.Equal) return t.value == taint.value;
  if (t.op == w.Exists) return true;
  if (t.op == w.Lt || t.op == w.Gt) {  // enableComparisonOperators == true at taints.go:89
    long long tv, xv;
    if (!go_atoi(str(t.value), tv) || !go_atoi(str(taint.value), xv)) return false;
    return t.op == w.Lt ? xv < tv : xv > tv;
  }
  return false;
}
// Taints.Tolerates — taints.go:83-95
inline bool taints_tolerated(const std::vector<Taint>& taints, const std::vector<Toleration>& tolerations) {
  for (auto& taint : taints) {
    bool ok = false;
    for (auto& t : tolerations) ok = ok || tolerates_taint(t, taint);
    if (!ok) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// metav1.LabelSelector -> labels.Selector (LabelSelectorAsSelector): nil selector matches nothing,
// empty selector matches everything (topologygroup.go:101-104, :443).
// ---------------------------------------------------------------------------------------------------------------
enum class SelOp { In, NotIn, Exists, DoesNotExist, Invalid };
struct SelectorExpr {
  Sym key = kNoSym;
  SelOp op = SelOp::Invalid;
  Sym op_text = kNoSym;   // the operator as written (an unknown one makes the selector invalid, but still tells groups apart)
  SymSet values;
  bool operator==(const SelectorExpr& o) const { return key == o.key && op_text == o.op_text && values == o.values; }
};
struct LabelSelector {
  bool is_nil = true;
  SymMap match_labels;
  std::vector<SelectorExpr> match_expressions;

  bool valid() const {
    for (auto& e : match_expressions) {
      if (e.op == SelOp::In || e.op == SelOp::NotIn) { if (e.values.empty()) return false; }
      else if (e.op == SelOp::Exists || e.op == SelOp::DoesNotExist) { if (!e.values.empty()) return false; }
      else return false;
    }
    return true;
  }
  bool matches(const SymMap& labels) const {
    if (is_nil) return false;   // labels.Nothing()
    if (!valid()) return false; // parse error => labels.Nothing()  (topologygroup.go:102-104)
    for (auto& kv : match_labels) {
      auto* it = labels.find(kv.first);
      if (!it || it->second != kv.second) return false;
    }
    for (auto& e : match_expressions) {
      auto* it = labels.find(e.key);
      if (e.op == SelOp::In) { if (!it || !e.values.count(it->second)) return false; }
      else if (e.op == SelOp::NotIn) { if (it && e.values.count(it->second)) return false; }
      else if (e.op == SelOp::Exists) { if (!it) return false; }
      else if (e.op == SelOp::DoesNotExist) { if (it) return false; }
    }
    return true;
  }
  // the selector as a set of expressions (hashstructure hashes slices as sets, topologygroup.go:188-205)
  bool same_as(const LabelSelector& o) const {
    if (is_nil != o.is_nil || !(match_labels == o.match_labels)) return false;
    for (auto& e : match_expressions) { bool f = false; for (auto& x : o.match_expressions) f = f || e == x; if (!f) return false; }
    for (auto& e : o.match_expressions) { bool f = false; for (auto& x : match_expressions) f = f || e == x; if (!f) return false; }
    return true;
  }
};

}  // namespace oracle
