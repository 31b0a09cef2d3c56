// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Restatement of pkg/controllers/provisioning/scheduling/{topology.go, topologygroup.go, topologynodefilter.go,
// topologydomaingroup.go}. Where the reference iterates a Go map (topologygroup.go:259,274,355,372,380,432) the
// result of a tie is undefined in the reference itself; the oracle canonicalises every such choice to the
// lexicographically smallest domain (documented in DESIGN.md) — domains are kept in std::map for that reason.
#pragma once
#include "model.hpp"

namespace oracle {

enum class TopologyType { Spread = 0, PodAffinity = 1, PodAntiAffinity = 2 };

// TopologyNodeFilter — topologynodefilter.go:31-96
struct TopologyNodeFilter {
  std::vector<Requirements> requirements;
  std::string taint_policy, affinity_policy;  // "" for the zero value used by (anti-)affinity groups
  std::vector<Toleration> tolerations;

  // MakeTopologyNodeFilter — topologynodefilter.go:38-65
  static TopologyNodeFilter make(const Pod& p, const std::string& taint_policy, const std::string& affinity_policy) {
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
    if (requirements.empty() || affinity_policy == "Ignore") return true;
    for (auto& r : requirements) if (reqs.compatible(r, false)) return true;
    return false;
  }
  // Matches — topologynodefilter.go:68-79
  bool matches(const std::vector<Taint>& taints, const Requirements& reqs) const {
    bool a = true, t = true;
    if (affinity_policy == "Honor") a = matches_requirements(reqs);
    if (taint_policy == "Honor") t = taints_tolerated(taints, tolerations);
    return a && t;
  }
  // What hashstructure sees of this struct (topologygroup.go:188-205): unexported Requirement fields
  // (complement/values/gte/lte) are skipped by the hasher, so only (key, minValues) per requirement contribute;
  // slices are hashed as sets. Two filters that differ only in selector VALUES therefore share one group.
  typedef std::set<std::set<std::pair<std::string, int>>> ReqView;
  typedef std::set<std::tuple<std::string, std::string, std::string, std::string>> TolView;
  std::tuple<ReqView, std::string, std::string, TolView> hash_view() const {
    ReqView rv;
    for (auto& r : requirements) {
      std::set<std::pair<std::string, int>> one;
      for (auto& kv : r.m) one.insert({kv.first, kv.second.min_values ? *kv.second.min_values : -1});
      rv.insert(one);
    }
    TolView tv;
    for (auto& t : tolerations) tv.insert({t.key, t.op, t.value, t.effect});
    return {rv, taint_policy, affinity_policy, tv};
  }
};

// TopologyDomainGroup — topologydomaingroup.go:28-72
struct TopologyDomainGroup {
  std::map<std::string, std::vector<std::vector<Taint>>> d;
  void insert(const std::string& domain, const std::vector<Taint>& taints) {
    auto it = d.find(domain);
    if (it == d.end() || taints.empty()) { d[domain] = {taints}; return; }
    if (it->second[0].empty()) return;
    it->second.push_back(taints);
  }
  template <class F>
  void for_each_domain(const Pod& p, const std::string& taint_policy, F f) const {
    for (auto& kv : d) {
      if (taint_policy == "Ignore") { f(kv.first); continue; }
      for (auto& taints : kv.second) if (taints_tolerated(taints, p.tolerations)) { f(kv.first); break; }
    }
  }
};

// TopologyGroup — topologygroup.go:55-126
struct TopologyGroup {
  std::string key;
  TopologyType type = TopologyType::Spread;
  int max_skew = 0;
  std::optional<int> min_domains;
  std::set<std::string> namespaces;
  LabelSelector selector;
  TopologyNodeFilter node_filter;
  std::set<std::string> owners;
  std::map<std::string, int> domains;
  std::set<std::string> empty_domains;

  static TopologyGroup make(TopologyType type, const std::string& key, const Pod& pod, const std::set<std::string>& namespaces,
                            const LabelSelector& sel, int max_skew, std::optional<int> min_domains,
                            const std::optional<std::string>& taint_policy, const std::optional<std::string>& affinity_policy,
                            const TopologyDomainGroup& dg) {
    TopologyGroup g;
    g.type = type; g.key = key; g.namespaces = namespaces; g.selector = sel; g.max_skew = max_skew; g.min_domains = min_domains;
    if (type == TopologyType::Spread) {
      std::string tp = taint_policy ? *taint_policy : "Ignore";
      std::string ap = affinity_policy ? *affinity_policy : "Honor";
      g.node_filter = TopologyNodeFilter::make(pod, tp, ap);
    }
    dg.for_each_domain(pod, g.node_filter.taint_policy, [&](const std::string& dom) { g.domains[dom] = 0; g.empty_domains.insert(dom); });
    return g;
  }
  // Hash() equivalence (topologygroup.go:188-222): same fields the hasher sees; minDomains is NOT hashed.
  typedef std::tuple<bool, std::map<std::string, std::string>, std::set<SelectorExpr>> SelView;
  auto identity() const {
    SelView sv{selector.is_nil, selector.match_labels, std::set<SelectorExpr>(selector.match_expressions.begin(), selector.match_expressions.end())};
    return std::make_tuple(key, (int)type, namespaces, max_skew, node_filter.hash_view(), sv);
  }
  bool selects(const Pod& p) const { return namespaces.count(p.ns) && selector.matches(p.labels); }  // :442
  bool counts(const Pod& p, const std::vector<Taint>& taints, const Requirements& reqs) const {      // :152
    return selects(p) && node_filter.matches(taints, reqs);
  }
  void record(const std::string& dom) { domains[dom]++; empty_domains.erase(dom); }                 // :143
  void reg(const std::string& dom) { if (!domains.count(dom)) { domains[dom] = 0; empty_domains.insert(dom); } }  // :157
  void unreg(const std::string& dom) { domains.erase(dom); empty_domains.erase(dom); }              // :166

  static Requirement dne(const std::string& key) { return Requirement::make(key, Op::DoesNotExist); }
  int domain_count(const std::string& d) const { auto it = domains.find(d); return it == domains.end() ? 0 : it->second; }

  // domainMinCount — topologygroup.go:300-322
  int domain_min_count(const Requirement& pod_domains) const {
    if (key == kLabelHostname) return 0;
    int mn = INT32_MAX, supported = 0;
    for (auto& kv : domains) if (pod_domains.has(kv.first)) { supported++; if (kv.second < mn) mn = kv.second; }
    if (min_domains && supported < *min_domains) mn = 0;
    return mn;
  }
  // nextDomainTopologySpread — topologygroup.go:229-298
  Requirement next_spread(const Pod& pod, const Requirement& pod_domains, const Requirement& node_domains, std::set<std::string>* valid = nullptr) const {
    int mn = domain_min_count(pod_domains);
    bool self = selects(pod);
    std::string min_domain;
    bool have = false;
    int min_count = INT32_MAX;
    if (key == kLabelHostname && node_domains.values.size() == 1) {
      const std::string& host = *node_domains.values.begin();
      int count = domain_count(host);
      if (self) count++;
      if (count <= max_skew) { if (valid) valid->insert(host); return Requirement::make(key, Op::In, {host}); }
      return dne(key);
    }
    auto consider = [&](const std::string& dom, int count) {
      if (self) count++;
      // int32 arithmetic as in the reference: count-min with min==MaxInt32 (no supported domain) stays negative
      if ((long long)count - (long long)mn <= (long long)max_skew) {
        if (valid) valid->insert(dom);
        if (count < min_count) { min_domain = dom; min_count = count; have = true; }  // first strict minimum in sorted order
      }
    };
    if (node_domains.op() == Op::In) {
      for (auto& dom : node_domains.values) { auto it = domains.find(dom); if (it != domains.end()) consider(dom, it->second); }
    } else {
      for (auto& kv : domains) if (node_domains.has(kv.first)) consider(kv.first, kv.second);
    }
    if (!have || min_domain.empty()) return dne(key);
    return Requirement::make(key, Op::In, {min_domain});
  }
  bool any_compatible_pod_domain(const Requirement& pod_domains) const {  // :393-400
    for (auto& kv : domains) if (pod_domains.has(kv.first) && kv.second > 0) return true;
    return false;
  }
  // nextDomainAffinity — topologygroup.go:324-388
  Requirement next_affinity(const Pod& pod, const Requirement& pod_domains, const Requirement& node_domains) const {
    Requirement options = dne(key);
    if (key == kLabelHostname && node_domains.values.size() == 1) {
      const std::string& host = *node_domains.values.begin();
      if (!pod_domains.has(host)) return options;
      if (domain_count(host) > 0) { options.values.insert(host); return options; }
      if (selects(pod) && (domains.size() == empty_domains.size() || !any_compatible_pod_domain(pod_domains))) { options.values.insert(host); return options; }
      return options;
    }
    if (node_domains.op() == Op::In) {
      for (auto& dom : node_domains.values) { auto it = domains.find(dom); if (pod_domains.has(dom) && it != domains.end() && it->second > 0) options.values.insert(dom); }
    } else {
      for (auto& kv : domains) if (pod_domains.has(kv.first) && kv.second > 0 && node_domains.has(kv.first)) options.values.insert(kv.first);
    }
    if (options.len() != 0) return options;
    if (selects(pod) && (domains.size() == empty_domains.size() || !any_compatible_pod_domain(pod_domains))) {
      Requirement inter = pod_domains.intersection(node_domains);
      for (auto& kv : domains) if (inter.has(kv.first)) { options.values.insert(kv.first); break; }  // canonical: smallest
      for (auto& kv : domains) if (pod_domains.has(kv.first)) { options.values.insert(kv.first); break; }
    }
    return options;
  }
  // nextDomainAntiAffinity — topologygroup.go:404-439
  Requirement next_anti_affinity(const Requirement& pod_domains, const Requirement& node_domains) const {
    Requirement options = dne(key);
    if (key == kLabelHostname && node_domains.values.size() == 1) {
      const std::string& host = *node_domains.values.begin();
      if (domain_count(host) == 0) options.values.insert(host);
      return options;
    }
    if (node_domains.op() == Op::In && node_domains.len() < (long long)empty_domains.size()) {
      for (auto& dom : node_domains.values) if (empty_domains.count(dom) && pod_domains.has(dom)) options.values.insert(dom);
    } else {
      for (auto& dom : empty_domains) if (node_domains.has(dom) && pod_domains.has(dom)) options.values.insert(dom);
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
  // insertion-ordered; identity() stands in for Hash() (topology.go:181-191)
  std::vector<TopologyGroup> groups, inverse_groups;
  std::map<std::string, TopologyDomainGroup> domain_groups;
  std::set<std::string> excluded_pods;
  const Problem* problem = nullptr;
  std::vector<const StateNode*> state_nodes;

  // buildDomainGroups — topology.go:105-146
  static std::map<std::string, TopologyDomainGroup> build_domain_groups(const Problem& pr, const std::vector<const NodePool*>& pools) {
    std::map<std::string, TopologyDomainGroup> dg;
    for (auto* np : pools) {
      for (int idx : np->instance_types) {
        const InstanceType& it = pr.catalog[idx];
        Requirements r = exprs_to_requirements(np->requirements);
        r.add_all(label_requirements(np->labels));
        r.add_all(it.reqs);
        for (auto& kv : r.m) for (auto& dom : kv.second.values) dg[kv.first].insert(dom, np->taints);
      }
      Requirements r = exprs_to_requirements(np->requirements);
      r.add_all(label_requirements(np->labels));
      for (auto& kv : r.m) if (kv.second.op() == Op::In) for (auto& v : kv.second.values) dg[kv.first].insert(v, np->taints);
    }
    return dg;
  }
  const TopologyDomainGroup& dgroup(const std::string& key) { return domain_groups[key]; }

  // buildNamespaceList — topology.go:536-557: the pod's namespace when the term names none; otherwise the listed
  // namespaces plus those whose labels the namespaceSelector matches (the problem's `namespaces` stand in for the lister)
  std::set<std::string> namespace_list(const std::string& ns, const PodAffinityTerm& term) const {
    if (term.namespaces.empty() && term.namespace_selector.is_nil) return {ns};
    std::set<std::string> out(term.namespaces.begin(), term.namespaces.end());
    if (!term.namespace_selector.is_nil && problem) for (auto& n : problem->namespaces) if (term.namespace_selector.matches(n.second)) out.insert(n.first);
    return out;
  }
  // updateInverseAntiAffinity — topology.go:329-355
  void update_inverse_anti_affinity(const Pod& pod, const std::map<std::string, std::string>* node_labels) {
    for (auto& term : pod.anti_required) {
      TopologyGroup tg = TopologyGroup::make(TopologyType::PodAntiAffinity, term.topology_key, pod, namespace_list(pod.ns, term),
                                             term.selector, INT32_MAX, std::nullopt, std::nullopt, std::nullopt, dgroup(term.topology_key));
      auto id = tg.identity();
      TopologyGroup* g = nullptr;
      for (auto& e : inverse_groups) if (e.identity() == id) { g = &e; break; }
      if (!g) { inverse_groups.push_back(tg); g = &inverse_groups.back(); }
      if (node_labels) { auto it = node_labels->find(g->key); if (it != node_labels->end()) g->record(it->second); }
      g->owners.insert(pod.uid);
    }
  }
  const StateNode* find_node(const std::string& name) const {
    for (auto* n : state_nodes) if (n->name == name) return n;   // the nodes of this simulation
    return nullptr;
  }
  // countDomains — topology.go:361-459 (kube reads replaced by the problem's clusterPods / stateNodes)
  void count_domains(TopologyGroup& tg) {
    for (auto* n : state_nodes) {
      if (!n->has_node) continue;
      if (!tg.node_filter.matches(n->taints, label_requirements(n->labels))) continue;
      auto it = n->labels.find(tg.key);
      if (it == n->labels.end()) continue;
      tg.reg(it->second);
    }
    for (auto& p : problem->cluster_pods) {
      if (!tg.namespaces.count(p.ns)) continue;
      if (!tg.selector.is_nil && !tg.selector.matches(p.labels)) continue;  // TopologyListOptions: nil selector lists everything
      if (p.node_name.empty() || p.phase == "Failed" || p.phase == "Succeeded") continue;  // IgnoredForTopology :614
      if (excluded_pods.count(p.uid)) continue;
      const StateNode* node = find_node(p.node_name);
      if (!node) continue;
      std::string dom;
      auto it = node->labels.find(tg.key);
      if (it != node->labels.end()) dom = it->second;
      else if (tg.key == kLabelHostname) dom = node->name;
      else continue;
      if (!tg.node_filter.matches(node->taints, label_requirements(node->labels))) continue;
      tg.record(dom);
    }
  }
  // newForTopologies — topology.go:461-495
  std::vector<TopologyGroup> new_for_topologies(Pod& p) {
    std::vector<TopologyGroup> out;
    for (auto& tsc : p.tscs) {
      if (ignore_preferences && tsc.when_unsatisfiable != "DoNotSchedule") continue;
      for (auto& k : tsc.match_label_keys) {
        auto it = p.labels.find(k);
        if (it != p.labels.end()) { tsc.selector.is_nil = false; tsc.selector.match_expressions.push_back({k, "In", {it->second}}); }
      }
      out.push_back(TopologyGroup::make(TopologyType::Spread, tsc.topology_key, p, {p.ns}, tsc.selector, tsc.max_skew, tsc.min_domains,
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
    for (auto& g : groups) g.owners.erase(p.uid);
    bool any_anti = p.has_pod_anti_affinity && (!p.anti_required.empty() || !p.anti_preferred.empty());
    bool req_anti = any_anti && !p.anti_required.empty();
    if ((ignore_preferences && req_anti) || (!ignore_preferences && any_anti)) update_inverse_anti_affinity(p, nullptr);
    std::vector<TopologyGroup> tgs = new_for_topologies(p);
    for (auto& g : new_for_affinities(p)) tgs.push_back(g);
    for (auto& tg : tgs) {
      auto id = tg.identity();
      TopologyGroup* g = nullptr;
      for (auto& e : groups) if (e.identity() == id) { g = &e; break; }
      if (!g) { count_domains(tg); groups.push_back(tg); g = &groups.back(); }
      g->owners.insert(p.uid);
    }
  }
  // NewTopology — topology.go:68-103
  void init(const Problem& pr, const std::vector<const NodePool*>& pools, const std::vector<const StateNode*>& snodes, std::vector<Pod>& pods, bool ignore_prefs) {
    problem = &pr;
    ignore_preferences = ignore_prefs;
    state_nodes = snodes;
    domain_groups = build_domain_groups(pr, pools);
    for (auto& p : pods) excluded_pods.insert(p.uid);
    // updateInverseAffinities — topology.go:310-324 : bound pods with required anti-affinity
    for (auto& cp : pr.cluster_pods) {
      if (!(cp.has_pod_anti_affinity && !cp.anti_required.empty())) continue;
      if (excluded_pods.count(cp.uid)) continue;
      const StateNode* node = find_node(cp.node_name);
      if (!node) continue;
      update_inverse_anti_affinity(cp, &node->labels);
    }
    for (auto& p : pods) update(p);
  }
  // Register / Unregister — topology.go:284-308
  void reg(const std::string& key, const std::string& dom) {
    for (auto& g : groups) if (g.key == key) g.reg(dom);
    for (auto& g : inverse_groups) if (g.key == key) g.reg(dom);
  }
  // AddRequirements — topology.go:226-250 ; returns false when some matching topology has no valid domain
  bool add_requirements(const Pod& p, const std::vector<Taint>& taints, const Requirements& pod_reqs, const Requirements& node_reqs, Requirements& out) const {
    out = node_reqs;
    auto apply = [&](const TopologyGroup& tg) {
      Requirement pod_domains = pod_reqs.has(tg.key) ? pod_reqs.get(tg.key) : Requirement::make(tg.key, Op::Exists);
      Requirement node_domains = node_reqs.has(tg.key) ? node_reqs.get(tg.key) : Requirement::make(tg.key, Op::Exists);
      Requirement d = tg.get(p, pod_domains, node_domains);
      if (d.len() == 0) return false;
      out.add(d);
      return true;
    };
    // getMatchingTopologies — topology.go:561-574
    for (auto& tg : groups) if (tg.owners.count(p.uid)) if (!apply(tg)) return false;
    for (auto& tg : inverse_groups) if (tg.counts(p, taints, node_reqs)) if (!apply(tg)) return false;
    return true;
  }
  // Record — topology.go:197-220
  void record(const Pod& p, const std::vector<Taint>& taints, const Requirements& reqs) {
    for (auto& tg : groups) {
      if (tg.counts(p, taints, reqs)) {
        Requirement domains = reqs.get(tg.key);
        if (tg.type == TopologyType::PodAntiAffinity) { for (auto& v : domains.values) tg.record(v); }
        else if (domains.len() == 1) tg.record(*domains.values.begin());
      }
    }
    for (auto& tg : inverse_groups) if (tg.owners.count(p.uid)) { Requirement d = reqs.get(tg.key); for (auto& v : d.values) tg.record(v); }
  }
};

}  // namespace oracle
