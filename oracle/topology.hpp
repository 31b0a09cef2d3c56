// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Restatement of pkg/controllers/provisioning/scheduling/{topology.go, topologygroup.go, topologynodefilter.go,
// topologydomaingroup.go}. Where the reference iterates a Go map (topologygroup.go:259,274,355,372,380,432) the
// result of a tie is undefined in the reference itself; the oracle canonicalises every such choice to the
// lexicographically smallest domain (documented in DESIGN.md) — domains are kept in a std::map ordered by the domain's
// STRING for that reason (DomainTable below; a hash index beside it answers the per-candidate lookups of the hostname
// fast paths, where a group holds one domain per NodeClaim).
#pragma once
#include <unordered_set>

#include "model.hpp"

namespace oracle {

enum class TopologyType { Spread = 0, PodAffinity = 1, PodAntiAffinity = 2 };

// TopologyNodeFilter — topologynodefilter.go:31-96
struct TopologyNodeFilter {
  std::vector<Requirements> requirements;
  Sym taint_policy = W().empty, affinity_policy = W().empty;  // "" for the zero value used by (anti-)affinity groups
  std::vector<Toleration> tolerations;

  // MakeTopologyNodeFilter — topologynodefilter.go:38-65
  static TopologyNodeFilter make(const Pod& p, Sym taint_policy, Sym affinity_policy) {
    TopologyNodeFilter f;
    f.taint_policy = taint_policy;
    f.affinity_policy = affinity_policy;
    f.tolerations = p.tolerations;
    Requirements sel = label_requirements(p.node_selector);
    if (!p.has_node_affinity || !p.has_required) { f.requirements.push_back(sel); return f; }
    for (auto& term : p.required_terms) {
      Requirements r;
      r.add_all(sel);
      r.add_all(exprs_to_requirements(term));
      f.requirements.push_back(r);
    }
    return f;
  }
  // matchesRequirements — topologynodefilter.go:84-96 (note: Matches does NOT forward compatibility options, :71)
  bool matches_requirements(const Requirements& reqs) const {
    if (requirements.empty() || affinity_policy == W().Ignore) return true;
    for (auto& r : requirements) if (reqs.compatible(r, false)) return true;
    return false;
  }
  // Matches — topologynodefilter.go:68-79
  bool matches(const std::vector<Taint>& taints, const Requirements& reqs) const {
    bool a = true, t = true;
    if (affinity_policy == W().Honor) a = matches_requirements(reqs);
    if (taint_policy == W().Honor) t = taints_tolerated(taints, tolerations);
    return a && t;
  }
  // What hashstructure sees of this struct (topologygroup.go:188-205): unexported Requirement fields
  // (complement/values/gte/lte) are skipped by the hasher, so only (key, minValues) per requirement contribute;
  // slices are hashed as sets. Two filters that differ only in selector VALUES therefore share one group.
  typedef std::set<std::set<std::pair<Sym, int>>> ReqView;
  typedef std::set<std::tuple<Sym, Sym, Sym, Sym>> TolView;
  std::tuple<ReqView, Sym, Sym, TolView> hash_view() const {
    ReqView rv;
    for (auto& r : requirements) {
      std::set<std::pair<Sym, int>> one;
      for (auto& q : r.m) one.insert({q.key, q.min_values ? *q.min_values : -1});
      rv.insert(one);
    }
    TolView tv;
    for (auto& t : tolerations) tv.insert({t.key, t.op, t.value, t.effect});
    return {rv, taint_policy, affinity_policy, tv};
  }
};

// TopologyDomainGroup — topologydomaingroup.go:28-72
struct TopologyDomainGroup {
  std::map<Sym, std::vector<std::vector<Taint>>> d;
  void insert(Sym domain, const std::vector<Taint>& taints) {
    auto it = d.find(domain);
    if (it == d.end() || taints.empty()) { d[domain] = {taints}; return; }
    if (it->second[0].empty()) return;
    it->second.push_back(taints);
  }
  template <class F>
  void for_each_domain(const Pod& p, Sym taint_policy, F f) const {
    for (auto& kv : d) {
      if (taint_policy == W().Ignore) { f(kv.first); continue; }
      for (auto& taints : kv.second) if (taints_tolerated(taints, p.tolerations)) { f(kv.first); break; }
    }
  }
};

// domains map[string]int32 + emptyDomains sets.Set[string] (topologygroup.go:68-70). `count` is the map; `names` holds its
// keys ordered by the domain's string (the canonical tie order — walked wherever the reference ranges over the map).
// Sym -> int32 with open addressing (one cache line per lookup; a hostname group holds one domain per NodeClaim and is asked
// about one of them by every candidate of every scan)
class CountMap {
  struct E { Sym k; int v; };
  std::vector<E> t_;
  size_t n_ = 0;
  static size_t h(Sym k) { return (size_t)((uint32_t)k * 2654435761u); }
  void grow() {
    std::vector<E> old;
    old.swap(t_);
    t_.assign(old.empty() ? 16 : old.size() * 2, E{kNoSym, 0});
    n_ = 0;
    for (auto& e : old) if (e.k != kNoSym) *slot(e.k, true) = e.v;
  }
  int* slot(Sym k, bool create) {
    if (t_.empty()) { if (!create) return nullptr; grow(); }
    const size_t mask = t_.size() - 1;
    for (size_t i = h(k) & mask;; i = (i + 1) & mask) {
      if (t_[i].k == k) return &t_[i].v;
      if (t_[i].k == kNoSym) {
        if (!create) return nullptr;
        if ((n_ + 1) * 10 > t_.size() * 7) { grow(); return slot(k, true); }
        t_[i].k = k; t_[i].v = 0; n_++;
        return &t_[i].v;
      }
    }
  }

 public:
  size_t size() const { return n_; }
  const int* find(Sym k) const { return const_cast<CountMap*>(this)->slot(k, false); }
  // (value, inserted)
  std::pair<int*, bool> emplace(Sym k) { const size_t before = n_; int* v = slot(k, true); return {v, n_ != before}; }
  void erase(Sym k) {   // rebuild without k (Unregister is not on the scheduling path)
    if (!find(k)) return;
    std::vector<E> old;
    old.swap(t_);
    t_.assign(old.size(), E{kNoSym, 0});
    n_ = 0;
    for (auto& e : old) if (e.k != kNoSym && e.k != k) *slot(e.k, true) = e.v;
  }
};
struct DomainTable {
  CountMap count;
  std::set<Sym, SymLexLess> names;
  std::set<Sym, SymLexLess> empty;   // emptyDomains
  size_t size() const { return count.size(); }
  bool has(Sym d) const { return count.find(d) != nullptr; }
  const int* find(Sym d) const { return count.find(d); }
  int of(Sym d) const { const int* c = count.find(d); return c ? *c : 0; }
  void put_zero(Sym d) { auto r = count.emplace(d); if (r.second) names.insert(d); *r.first = 0; empty.insert(d); }
  bool is_empty_domain(Sym d) const { const int* c = find(d); return c && *c == 0 && empty.count(d); }
};

// TopologyGroup — topologygroup.go:55-126
struct TopologyGroup {
  Sym key = kNoSym;
  TopologyType type = TopologyType::Spread;
  int max_skew = 0;
  std::optional<int> min_domains;
  SymSet namespaces;
  LabelSelector selector;
  TopologyNodeFilter node_filter;
  std::unordered_set<Sym> owners;
  DomainTable domains;

  static TopologyGroup make(TopologyType type, Sym key, const Pod& pod, const SymSet& namespaces,
                            const LabelSelector& sel, int max_skew, std::optional<int> min_domains,
                            const std::optional<Sym>& taint_policy, const std::optional<Sym>& affinity_policy,
                            const TopologyDomainGroup& dg) {
    TopologyGroup g;
    g.type = type; g.key = key; g.namespaces = namespaces; g.selector = sel; g.max_skew = max_skew; g.min_domains = min_domains;
    if (type == TopologyType::Spread) {
      Sym tp = taint_policy ? *taint_policy : W().Ignore;
      Sym ap = affinity_policy ? *affinity_policy : W().Honor;
      g.node_filter = TopologyNodeFilter::make(pod, tp, ap);
    }
    dg.for_each_domain(pod, g.node_filter.taint_policy, [&](Sym dom) { g.domains.put_zero(dom); });
    return g;
  }
  // Hash() equivalence (topologygroup.go:188-222): same fields the hasher sees; minDomains is NOT hashed.
  bool same_identity(const TopologyGroup& o) const {
    return key == o.key && type == o.type && namespaces == o.namespaces && max_skew == o.max_skew &&
           selector.same_as(o.selector) && node_filter.hash_view() == o.node_filter.hash_view();
  }
  bool selects(const Pod& p) const { return namespaces.count(p.ns) && selector.matches(p.labels); }  // :442
  bool counts(const Pod& p, const std::vector<Taint>& taints, const Requirements& reqs) const {      // :152
    return selects(p) && node_filter.matches(taints, reqs);
  }
  void record(Sym dom) {                                                                            // :143
    auto r = domains.count.emplace(dom);
    if (r.second) domains.names.insert(dom);
    if ((*r.first)++ == 0) domains.empty.erase(dom);   // emptyDomains.Delete: only a domain at zero can be in it
  }
  void reg(Sym dom) { if (!domains.has(dom)) domains.put_zero(dom); }                               // :157
  void unreg(Sym dom) { domains.count.erase(dom); domains.names.erase(dom); domains.empty.erase(dom); }   // :166

  static Requirement dne(Sym key) { return Requirement::make(key, Op::DoesNotExist); }
  int domain_count(Sym d) const { const int* c = domains.find(d); return c ? *c : 0; }

  // domainMinCount — topologygroup.go:300-322
  int domain_min_count(const Requirement& pod_domains) const {
    if (key == W().hostname) return 0;
    int mn = INT32_MAX, supported = 0;
    for (Sym d : domains.names) if (pod_domains.has(d)) { supported++; int c = domains.of(d); if (c < mn) mn = c; }
    if (min_domains && supported < *min_domains) mn = 0;
    return mn;
  }
  // nextDomainTopologySpread — topologygroup.go:229-298
  Requirement next_spread(const Pod& pod, const Requirement& pod_domains, const Requirement& node_domains) const {
    int mn = domain_min_count(pod_domains);
    bool self = selects(pod);
    Sym min_domain = kNoSym;
    bool have = false;
    int min_count = INT32_MAX;
    if (key == W().hostname && node_domains.values.size() == 1) {
      Sym host = *node_domains.values.begin();
      int count = domain_count(host);
      if (self) count++;
      if (count <= max_skew) return Requirement::make(key, Op::In, host);
      return dne(key);
    }
    auto consider = [&](Sym dom, int count) {
      if (self) count++;
      // int32 arithmetic as in the reference: count-min with min==MaxInt32 (no supported domain) stays negative
      if ((long long)count - (long long)mn <= (long long)max_skew) {
        if (count < min_count) { min_domain = dom; min_count = count; have = true; }  // first strict minimum in sorted order
      }
    };
    if (node_domains.op() == Op::In) {
      // the values in string order, first strict minimum == the minimum by (count, string): no need to sort them first
      for (Sym dom : node_domains.values) {
        const int* c = domains.find(dom);
        if (!c) continue;
        int count = *c + (self ? 1 : 0);
        if ((long long)count - (long long)mn > (long long)max_skew) continue;
        if (!have || count < min_count || (count == min_count && sym_lex_less(dom, min_domain))) { min_domain = dom; min_count = count; have = true; }
      }
    } else {
      for (Sym d : domains.names) if (node_domains.has(d)) consider(d, domains.of(d));
    }
    if (!have || min_domain == W().empty) return dne(key);
    return Requirement::make(key, Op::In, min_domain);
  }
  bool any_compatible_pod_domain(const Requirement& pod_domains) const {  // :393-400
    for (Sym d : domains.names) if (pod_domains.has(d) && domains.of(d) > 0) return true;
    return false;
  }
  // nextDomainAffinity — topologygroup.go:324-388
  Requirement next_affinity(const Pod& pod, const Requirement& pod_domains, const Requirement& node_domains) const {
    Requirement options = dne(key);
    if (key == W().hostname && node_domains.values.size() == 1) {
      Sym host = *node_domains.values.begin();
      if (!pod_domains.has(host)) return options;
      if (domain_count(host) > 0) { options.values.insert(host); return options; }
      if (selects(pod) && (domains.size() == domains.empty.size() || !any_compatible_pod_domain(pod_domains))) { options.values.insert(host); return options; }
      return options;
    }
    if (node_domains.op() == Op::In) {
      for (Sym dom : node_domains.values) { const int* c = domains.find(dom); if (pod_domains.has(dom) && c && *c > 0) options.values.insert(dom); }
    } else {
      for (Sym d : domains.names) if (pod_domains.has(d) && domains.of(d) > 0 && node_domains.has(d)) options.values.insert(d);
    }
    if (options.len() != 0) return options;
    if (selects(pod) && (domains.size() == domains.empty.size() || !any_compatible_pod_domain(pod_domains))) {
      Requirement inter = pod_domains.intersection(node_domains);
      for (Sym d : domains.names) if (inter.has(d)) { options.values.insert(d); break; }  // canonical: smallest
      for (Sym d : domains.names) if (pod_domains.has(d)) { options.values.insert(d); break; }
    }
    return options;
  }
  // nextDomainAntiAffinity — topologygroup.go:404-439
  Requirement next_anti_affinity(const Requirement& pod_domains, const Requirement& node_domains) const {
    Requirement options = dne(key);
    if (key == W().hostname && node_domains.values.size() == 1) {
      Sym host = *node_domains.values.begin();
      if (domain_count(host) == 0) options.values.insert(host);
      return options;
    }
    if (node_domains.op() == Op::In && node_domains.len() < (long long)domains.empty.size()) {
      for (Sym dom : node_domains.values) if (domains.is_empty_domain(dom) && pod_domains.has(dom)) options.values.insert(dom);
    } else {
      for (Sym dom : domains.empty) if (node_domains.has(dom) && pod_domains.has(dom)) options.values.insert(dom);
    }
    return options;
  }
  // Get — topologygroup.go:128-141
  Requirement get(const Pod& pod, const Requirement& pod_domains, const Requirement& node_domains) const {
    switch (type) {
      case TopologyType::Spread: return next_spread(pod, pod_domains, node_domains);
      case TopologyType::PodAffinity: return next_affinity(pod, pod_domains, node_domains);
      default: return next_anti_affinity(pod_domains, node_domains);
    }
  }
};

struct Topology {
  bool ignore_preferences = false;
  // insertion-ordered; same_identity() stands in for Hash() (topology.go:181-191)
  std::vector<TopologyGroup> groups, inverse_groups;
  // which of `groups` hold a pod as an owner (ascending positions): getMatchingTopologies (topology.go:561-574) walks every
  // group and asks owners.Has(uid); this index answers the same question for all groups at once. Kept in step with `owners`.
  std::unordered_map<Sym, std::vector<int>> owned;
  std::map<Sym, TopologyDomainGroup> domain_groups;
  std::unordered_set<Sym> excluded_pods;
  const Problem* problem = nullptr;
  std::vector<const StateNode*> state_nodes;
  std::unordered_map<Sym, const StateNode*> node_by_name;
  std::unordered_map<Sym, size_t> node_index;   // name -> position in state_nodes

  // buildDomainGroups — topology.go:105-146
  static std::map<Sym, TopologyDomainGroup> build_domain_groups(const Problem& pr, const std::vector<const NodePool*>& pools) {
    std::map<Sym, TopologyDomainGroup> dg;
    for (auto* np : pools) {
      for (int idx : np->instance_types) {
        const InstanceType& it = pr.catalog[idx];
        Requirements r = exprs_to_requirements(np->requirements);
        r.add_all(label_requirements(np->labels));
        r.add_all(it.reqs);
        for (auto& q : r.m) for (Sym dom : q.values) dg[q.key].insert(dom, np->taints);
      }
      Requirements r = exprs_to_requirements(np->requirements);
      r.add_all(label_requirements(np->labels));
      for (auto& q : r.m) if (q.op() == Op::In) for (Sym v : q.values) dg[q.key].insert(v, np->taints);
    }
    return dg;
  }
  const TopologyDomainGroup& dgroup(Sym key) { return domain_groups[key]; }

  // buildNamespaceList — topology.go:536-557: the pod's namespace when the term names none; otherwise the listed
  // namespaces plus those whose labels the namespaceSelector matches (the problem's `namespaces` stand in for the lister)
  SymSet namespace_list(Sym ns, const PodAffinityTerm& term) const {
    if (term.namespaces.empty() && term.namespace_selector.is_nil) return SymSet{ns};
    SymSet out;
    out.insert(term.namespaces.begin(), term.namespaces.end());
    if (!term.namespace_selector.is_nil && problem) for (auto& n : problem->namespaces) if (term.namespace_selector.matches(n.second)) out.insert(n.first);
    return out;
  }
  // updateInverseAntiAffinity — topology.go:329-355
  void update_inverse_anti_affinity(const Pod& pod, const SymMap* node_labels) {
    for (auto& term : pod.anti_required) {
      TopologyGroup tg = TopologyGroup::make(TopologyType::PodAntiAffinity, term.topology_key, pod, namespace_list(pod.ns, term),
                                             term.selector, INT32_MAX, std::nullopt, std::nullopt, std::nullopt, dgroup(term.topology_key));
      TopologyGroup* g = nullptr;
      for (auto& e : inverse_groups) if (e.same_identity(tg)) { g = &e; break; }
      if (!g) { inverse_groups.push_back(tg); g = &inverse_groups.back(); }
      if (node_labels) { auto* it = node_labels->find(g->key); if (it) g->record(it->second); }
      g->owners.insert(pod.uid_s);
    }
  }
  const StateNode* find_node(Sym name) const {
    auto it = node_by_name.find(name);   // the nodes of this simulation
    return it == node_by_name.end() ? nullptr : it->second;
  }
  // countDomains — topology.go:361-459 (kube reads replaced by the problem's clusterPods / stateNodes). The reference asks
  // TopologyNodeFilter.Matches(node) once for the node and once more for every pod bound to it; the answer is the node's, so it
  // is computed once per node here (a 100k-node cluster carries 2M bound pods, and every probe of a sweep counts every group).
  void count_domains(TopologyGroup& tg) {
    std::vector<char> node_matches(state_nodes.size(), 0);
    for (size_t i = 0; i < state_nodes.size(); ++i) {
      const StateNode* n = state_nodes[i];
      node_matches[i] = tg.node_filter.matches(n->taints, label_requirements(n->labels)) ? 1 : 0;
      if (!n->has_node) continue;
      if (!node_matches[i]) continue;
      auto* it = n->labels.find(tg.key);
      if (!it) continue;
      tg.reg(it->second);
    }
    for (auto& p : problem->cluster_pods) {
      if (!tg.namespaces.count(p.ns)) continue;
      if (!tg.selector.is_nil && !tg.selector.matches(p.labels)) continue;  // TopologyListOptions: nil selector lists everything
      if (p.node_name == W().empty || p.phase == W().Failed || p.phase == W().Succeeded) continue;  // IgnoredForTopology :614
      if (excluded_pods.count(p.uid_s)) continue;
      auto ni = node_index.find(p.node_name);
      if (ni == node_index.end()) continue;
      const StateNode* node = state_nodes[ni->second];
      Sym dom;
      auto* it = node->labels.find(tg.key);
      if (it) dom = it->second;
      else if (tg.key == W().hostname) dom = node->name_s;
      else continue;
      if (!node_matches[ni->second]) continue;
      tg.record(dom);
    }
  }
  // newForTopologies — topology.go:461-495
  std::vector<TopologyGroup> new_for_topologies(Pod& p) {
    std::vector<TopologyGroup> out;
    for (auto& tsc : p.tscs) {
      if (ignore_preferences && tsc.when_unsatisfiable != W().DoNotSchedule) continue;
      for (Sym k : tsc.match_label_keys) {
        auto* it = p.labels.find(k);
        if (it) {
          tsc.selector.is_nil = false;
          SelectorExpr x;
          x.key = k; x.op = SelOp::In; x.op_text = sym("In"); x.values.insert(it->second);
          tsc.selector.match_expressions.push_back(x);
        }
      }
      out.push_back(TopologyGroup::make(TopologyType::Spread, tsc.topology_key, p, SymSet{p.ns}, tsc.selector, tsc.max_skew, tsc.min_domains,
                                        tsc.node_taints_policy, tsc.node_affinity_policy, dgroup(tsc.topology_key)));
    }
    return out;
  }
  // newForAffinities — topology.go:498-538
  std::vector<TopologyGroup> new_for_affinities(const Pod& p) {
    std::vector<TopologyGroup> out;
    auto add = [&](TopologyType t, const PodAffinityTerm& term) {
      out.push_back(TopologyGroup::make(t, term.topology_key, p, namespace_list(p.ns, term), term.selector, INT32_MAX,
                                        std::nullopt, std::nullopt, std::nullopt, dgroup(term.topology_key)));
    };
    if (p.has_pod_affinity) {
      for (auto& t : p.affinity_required) add(TopologyType::PodAffinity, t);
      if (!ignore_preferences) for (auto& t : p.affinity_preferred) add(TopologyType::PodAffinity, t.term);
    }
    if (p.has_pod_anti_affinity) {
      for (auto& t : p.anti_required) add(TopologyType::PodAntiAffinity, t);
      if (!ignore_preferences) for (auto& t : p.anti_preferred) add(TopologyType::PodAntiAffinity, t.term);
    }
    return out;
  }
  // Update — topology.go:162-194
  void update(Pod& p) {
    auto ow = owned.find(p.uid_s);   // for _, tg := range t.topologyGroups { tg.RemoveOwner(p.UID) }: only these hold it
    if (ow != owned.end()) { for (int gi : ow->second) groups[gi].owners.erase(p.uid_s); owned.erase(ow); }
    bool any_anti = p.has_pod_anti_affinity && (!p.anti_required.empty() || !p.anti_preferred.empty());
    bool req_anti = any_anti && !p.anti_required.empty();
    if ((ignore_preferences && req_anti) || (!ignore_preferences && any_anti)) update_inverse_anti_affinity(p, nullptr);
    std::vector<TopologyGroup> tgs = new_for_topologies(p);
    for (auto& g : new_for_affinities(p)) tgs.push_back(g);
    for (auto& tg : tgs) {
      int gi = -1;
      for (size_t i = 0; i < groups.size(); ++i) if (groups[i].same_identity(tg)) { gi = (int)i; break; }
      if (gi < 0) { count_domains(tg); groups.push_back(tg); gi = (int)groups.size() - 1; }
      if (groups[gi].owners.insert(p.uid_s).second) {
        std::vector<int>& l = owned[p.uid_s];
        l.insert(std::upper_bound(l.begin(), l.end(), gi), gi);
      }
    }
  }
  // NewTopology — topology.go:68-103
  void init(const Problem& pr, const std::vector<const NodePool*>& pools, const std::vector<const StateNode*>& snodes, std::vector<Pod>& pods, bool ignore_prefs) {
    problem = &pr;
    ignore_preferences = ignore_prefs;
    state_nodes = snodes;
    for (size_t i = 0; i < snodes.size(); ++i) { node_by_name.emplace(snodes[i]->name_s, snodes[i]); node_index.emplace(snodes[i]->name_s, i); }   // first of equal names, as the linear search found
    domain_groups = build_domain_groups(pr, pools);
    for (auto& p : pods) excluded_pods.insert(p.uid_s);
    // updateInverseAffinities — topology.go:310-324 : bound pods with required anti-affinity
    for (auto& cp : pr.cluster_pods) {
      if (!(cp.has_pod_anti_affinity && !cp.anti_required.empty())) continue;
      if (excluded_pods.count(cp.uid_s)) continue;
      const StateNode* node = find_node(cp.node_name);
      if (!node) continue;
      update_inverse_anti_affinity(cp, &node->labels);
    }
    for (auto& p : pods) update(p);
  }
  // Register / Unregister — topology.go:284-308
  void reg(Sym key, Sym dom) {
    for (auto& g : groups) if (g.key == key) g.reg(dom);
    for (auto& g : inverse_groups) if (g.key == key) g.reg(dom);
  }
  // AddRequirements — topology.go:226-250 ; returns false when some matching topology has no valid domain. `out` is built
  // only once every matching topology has produced a domain (the reference copies nodeRequirements first, :227, and returns
  // an error before anyone reads the copy; node_domains is read from nodeRequirements, never from the copy).
  bool add_requirements(const Pod& p, const std::vector<Taint>& taints, const Requirements& pod_reqs, const Requirements& node_reqs, Requirements& out) const {
    Requirement adds_inl[4];
    std::vector<Requirement> adds_more;
    size_t n_adds = 0;
    auto apply = [&](const TopologyGroup& tg) {
      const Requirement* pr_ = pod_reqs.find(tg.key);
      const Requirement* nr_ = node_reqs.find(tg.key);
      Requirement pod_exists, node_exists;   // NewRequirement(topology.Key, Exists) when the set does not have the key
      if (!pr_) { pod_exists = Requirement::make(tg.key, Op::Exists); pr_ = &pod_exists; }
      if (!nr_) { node_exists = Requirement::make(tg.key, Op::Exists); nr_ = &node_exists; }
      Requirement d = tg.get(p, *pr_, *nr_);
      if (d.len() == 0) return false;
      if (n_adds < 4) adds_inl[n_adds] = std::move(d); else adds_more.push_back(std::move(d));
      n_adds++;
      return true;
    };
    // getMatchingTopologies — topology.go:561-574
    auto ow = owned.find(p.uid_s);
    if (ow != owned.end()) for (int gi : ow->second) if (!apply(groups[gi])) return false;
    for (auto& tg : inverse_groups) if (tg.counts(p, taints, node_reqs)) if (!apply(tg)) return false;
    out = node_reqs;
    for (size_t i = 0; i < n_adds; ++i) out.add(i < 4 ? adds_inl[i] : adds_more[i - 4]);
    return true;
  }
  // Record — topology.go:197-220
  void record(const Pod& p, const std::vector<Taint>& taints, const Requirements& reqs) {
    for (auto& tg : groups) {
      if (tg.counts(p, taints, reqs)) {
        Requirement domains = reqs.get(tg.key);
        if (tg.type == TopologyType::PodAntiAffinity) { for (Sym v : domains.values) tg.record(v); }
        else if (domains.len() == 1) tg.record(*domains.values.begin());
      }
    }
    for (auto& tg : inverse_groups) if (tg.owners.count(p.uid_s)) { Requirement d = reqs.get(tg.key); for (Sym v : d.values) tg.record(v); }
  }
};

}  // namespace oracle
