// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Restatement of pkg/controllers/provisioning/scheduling/{scheduler.go, nodeclaim.go, existingnode.go, queue.go,
// nodeclaimtemplate.go, preferences.go, reservationmanager.go} and pkg/cloudprovider/types.go (price/minValues helpers).
// Single-threaded: parallelizeUntil (scheduler.go:939-961) only fans out candidate checks and always keeps the
// lowest index (:639,:674,:759), so a sequential first-success scan is equivalent.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include "topology.hpp"

namespace oracle {

// ---- cloudprovider helpers -------------------------------------------------------------------------------------
// Offerings.HasCompatible / Compatible — types.go:553-570
inline bool offering_compatible(const Requirements& reqs, const Offering& o) { return reqs.compatible(o.reqs, true); }

// InstanceTypes.SatisfiesMinValues — types.go:399-433
inline int satisfies_min_values(const std::vector<const InstanceType*>& its, const Requirements& reqs, std::map<Sym, int>* unsat, bool* ok) {
  *ok = true;
  if (!reqs.has_min_values()) return 0;
  std::map<Sym, int> incompatible;
  std::map<Sym, SymSet> values_for_key;
  for (size_t i = 0; i < its.size(); ++i) {
    for (auto& q : reqs.m) if (q.min_values) {
      Requirement r = its[i]->reqs.get(q.key);
      values_for_key[q.key].insert(r.values.begin(), r.values.end());
    }
    for (auto& kv : values_for_key) {
      int mv = *reqs.get(kv.first).min_values;
      if ((int)kv.second.size() < mv) incompatible[kv.first] = (int)kv.second.size();
      else incompatible.erase(kv.first);
    }
    if (incompatible.empty()) return (int)i + 1;
  }
  if (!incompatible.empty()) { *ok = false; if (unsat) *unsat = incompatible; return (int)its.size(); }
  return (int)its.size();
}
// cheapest compatible available offering price (the comparator key of OrderByPrice, types.go:336-355)
inline double min_compatible_price(const InstanceType& it, const Requirements& reqs) {
  double p = DBL_MAX;
  for (auto& o : it.offerings) if (o.available && offering_compatible(reqs, o) && o.price < p) p = o.price;
  return p;
}
// InstanceTypes.OrderByPrice — types.go:336-355 (sort.Slice, unstable; Go pdqsort restated in pdqsort.hpp)
inline void order_by_price(std::vector<const InstanceType*>& its, const Requirements& reqs) {
  go_sort_slice(its, [&](const InstanceType* a, const InstanceType* b) { return min_compatible_price(*a, reqs) < min_compatible_price(*b, reqs); });
}
// Offerings.WorstLaunchPrice — types.go:587-598
inline double worst_launch_price(const InstanceType& it, const Requirements& reqs) {
  for (Sym ct : {W().reserved, W().spot, W().on_demand}) {
    Requirements ctr;
    ctr.add(Requirement::make(W().capacity_type, Op::In, ct));
    double worst = -1;
    bool any = false;
    for (auto& o : it.offerings) {
      if (!o.available) continue;
      if (!offering_compatible(reqs, o)) continue;
      if (!ctr.compatible(o.reqs, true)) continue;
      if (!any || o.price > worst) worst = o.price;  // MostExpensive: lo.MaxBy keeps the first maximum
      any = true;
    }
    if (any) return worst;
  }
  return DBL_MAX;
}

// ---- ReservationManager — reservationmanager.go:28-110 ---------------------------------------------------------
struct ReservationManager {
  std::map<Sym, SymSet> reservations;  // hostname -> reservation ids
  std::map<Sym, int> capacity;         // reservation id -> remaining
  void init(const Problem& pr, const std::vector<const NodePool*>& pools) {
    for (auto* np : pools)
      for (int idx : np->instance_types)
        for (auto& o : pr.catalog[idx].offerings) {
          if (o.capacity_type() != W().reserved) continue;
          Sym id = o.reservation_id();
          auto it = capacity.find(id);
          if (it == capacity.end()) capacity[id] = o.reservation_capacity;
          else if (o.reservation_capacity < it->second) it->second = o.reservation_capacity;  // keep the most pessimistic
        }
  }
  bool has_reservation(Sym host, Sym id) const {
    auto it = reservations.find(host);
    return it != reservations.end() && it->second.count(id);
  }
  bool can_reserve(Sym host, const Offering& o) const {
    Sym id = o.reservation_id();
    if (has_reservation(host, id)) return true;
    auto it = capacity.find(id);
    return it != capacity.end() && it->second > 0;
  }
  void reserve(Sym host, const std::vector<const Offering*>& ofs) {
    for (auto* o : ofs) {
      Sym id = o->reservation_id();
      if (has_reservation(host, id)) continue;
      capacity[id] -= 1;
      reservations[host].insert(id);
    }
  }
  void release(Sym host, const Offering& o) {
    Sym id = o.reservation_id();
    auto it = reservations.find(host);
    if (it != reservations.end() && it->second.count(id)) { it->second.erase(id); capacity[id] += 1; }
  }
};

// ---- scheduling types ------------------------------------------------------------------------------------------
struct DaemonOverheadGroup {  // scheduler.go:963-967
  std::set<const InstanceType*> its;
  std::vector<const InstanceType*> its_ordered;
  ResourceList overhead;
  std::vector<HostPort> host_ports;   // HostPortUsage of the group's daemon pods (scheduler.go:990-993)
};

// NodeClaimTemplate — nodeclaimtemplate.go:55-94
struct NodeClaimTemplate {
  const NodePool* np = nullptr;
  std::string nodepool_name;
  int weight = 0;
  std::vector<const InstanceType*> its;
  Requirements reqs;
  std::vector<Taint> taints;
  std::vector<DaemonOverheadGroup> daemon_groups;
  int index = 0;
};

enum ErrCode {
  ERR_NONE = 0,
  ERR_TAINTS = 1,            // did not tolerate taint (taints.go:91)
  ERR_INCOMPATIBLE = 2,      // incompatible requirements (nodeclaim.go:134)
  ERR_TOPOLOGY = 3,          // unsatisfiable topology constraint (topology.go:241)
  ERR_INSTANCE_TYPES = 4,    // InstanceTypeFilterError (nodeclaim.go:437)
  ERR_RESOURCES = 5,         // exceeds node resources (existingnode.go:97)
  ERR_NO_TEMPLATES = 6,      // nodepool requirements filtered out all available instance types (scheduler.go:605)
  ERR_LIMITS = 7,            // all available instance types exceed limits / node limits exhausted (scheduler.go:713,718)
  ERR_RESERVED = 8,          // ReservedOfferingError (nodeclaim.go:341,346)
  ERR_EXISTING = 9,          // failed scheduling pod to existing nodes
  ERR_MIN_VALUES = 10,       // minValues requirement is not met (types.go:430)
};

struct FilterDiag {  // InstanceTypeFilterError flags — nodeclaim.go:437-459
  bool requirements_met = false, fits = false, has_offering = false;
  bool requirements_and_fits = false, requirements_and_offering = false, fits_and_offering = false;
  bool min_values_incompatible = false;
  int bits() const {
    return (requirements_met ? 1 : 0) | (fits ? 2 : 0) | (has_offering ? 4 : 0) | (requirements_and_fits ? 8 : 0) |
           (requirements_and_offering ? 16 : 0) | (fits_and_offering ? 32 : 0) | (min_values_incompatible ? 64 : 0);
  }
};

struct Counters {
  long long bin_evaluations = 0;       // V: CanAdd calls on in-flight claims + new-claim attempts + existing nodes
  long long it_evaluations = 0;        // instance types visited in filterInstanceTypesByRequirements
  long long sorts = 0, pops = 0, relaxations = 0;
};

struct PodData {  // scheduler.go:217-229
  ResourceList requests;
  Requirements reqs, strict_reqs;
  std::vector<Requirements> volume_reqs;   // VolumeRequirements (scheduler.go:222, :572): alternatives, tried in order
};

// compatible / fits — nodeclaim.go:620-638
inline bool it_compatible(const InstanceType& it, const Requirements& reqs) { return it.reqs.intersects(reqs); }
inline void it_fits(const InstanceType& it, const ResourceList& requests, const Requirements& reqs, bool& fits, bool& has_offering) {
  fits = false; has_offering = false;
  for (auto& g : it.groups) {
    bool resource_fit = res_fits(requests, g.allocatable);
    for (auto* of : g.offerings) {
      if (offering_compatible(reqs, *of)) {
        has_offering = true;
        if (resource_fit) { fits = true; return; }
        break;
      }
    }
  }
}

// filterInstanceTypesByRequirements — nodeclaim.go:541-618
inline bool filter_instance_types(const std::vector<const InstanceType*>& its, Requirements& reqs, const std::vector<DaemonOverheadGroup>& groups,
                                  const ResourceList& total_requests, bool relax_min_values, std::vector<const InstanceType*>& remaining,
                                  std::map<Sym, int>& unsat, FilterDiag& d, Counters* ctr,
                                  const std::vector<HostPort>* pod_ports = nullptr, const std::vector<HostPort>* bin_ports = nullptr) {
  remaining.clear();
  // membership in `its` (the NodeClaim's InstanceTypeOptions) by catalogue position
  int max_index = -1;
  for (auto* it : its) max_index = std::max(max_index, it->catalog_index);
  std::vector<char> eligible((size_t)(max_index + 1), 0);
  for (auto* it : its) eligible[(size_t)it->catalog_index] = 1;
  for (auto& g : groups) {
    // a group whose host ports (its daemon pods' + the pods already on the NodeClaim, nodeclaim.go:256-259) collide with
    // the pod's is skipped as a whole (nodeclaim.go:562-565)
    if (pod_ports && !pod_ports->empty() && (host_ports_conflict(*pod_ports, g.host_ports) || (bin_ports && host_ports_conflict(*pod_ports, *bin_ports)))) continue;
    ResourceList total = g.overhead.empty() ? total_requests : res_merge(total_requests, g.overhead);
    for (auto* it : g.its_ordered) {
      if (it->catalog_index > max_index || !eligible[(size_t)it->catalog_index]) continue;
      if (ctr) ctr->it_evaluations++;
      bool compat = it_compatible(*it, reqs);
      bool fits, has_off;
      it_fits(*it, total, reqs, fits, has_off);
      d.requirements_met |= compat; d.fits |= fits; d.has_offering |= has_off;
      d.requirements_and_fits |= (compat && fits && !has_off);
      d.requirements_and_offering |= (compat && has_off && !fits);
      d.fits_and_offering |= (fits && has_off && !compat);
      if (compat && fits && has_off) remaining.push_back(it);
    }
  }
  if (reqs.has_min_values()) {
    bool ok;
    std::map<Sym, int> u;
    satisfies_min_values(remaining, reqs, &u, &ok);
    if (!ok) {
      unsat = u;
      if (!relax_min_values) { remaining.clear(); d.min_values_incompatible = true; }
    }
  }
  return !remaining.empty();
}

struct Scheduler;

// NodeClaim — nodeclaim.go:43-119
struct NodeClaim {
  const NodeClaimTemplate* tmpl = nullptr;
  Requirements reqs;
  std::vector<const InstanceType*> its;
  ResourceList requests;
  std::vector<Pod*> pods;
  Sym hostname = kNoSym;
  std::map<std::string, std::string> annotations;
  std::vector<const Offering*> reserved_offerings;
  std::vector<HostPort> host_ports;   // what Add put into every daemon group's HostPortUsage (nodeclaim.go:256-259)
  int id = 0;  // creation order
};

// ExistingNode — existingnode.go:32-75
struct ExistingNode {
  const StateNode* node = nullptr;
  std::vector<Pod*> pods;
  ResourceList remaining;
  Requirements reqs;
  std::vector<HostPort> host_ports;   // StateNode.HostPortUsage() + the pods added in this Solve (existingnode.go:178)
  std::map<Sym, SymSet> volumes;   // StateNode.VolumeUsage().volumes + the pods added in this Solve (existingnode.go:179)
  bool under_consolidate_after = false;
};

struct Results {
  std::vector<NodeClaim*> new_node_claims;
  std::vector<ExistingNode*> existing_nodes;
  std::map<std::string, std::pair<int, int>> pod_errors;  // uid -> (code, diag bits)
  bool timed_out = false;
};

struct Scheduler {
  const Problem* pr = nullptr;
  Options opts;
  std::vector<Pod> pods;          // working copies ("original" pods held by the queue)
  std::vector<const NodePool*> pools;
  std::vector<NodeClaimTemplate> templates;
  std::vector<std::unique_ptr<NodeClaim>> claim_store;
  std::vector<NodeClaim*> new_node_claims;   // s.newNodeClaims: physically re-sorted in place (scheduler.go:598)
  std::vector<uint32_t> claim_pods;          // len(new_node_claims[i].Pods), moved with it: the sort's less() reads it here instead of
                                             // chasing 200k NodeClaim pointers per sort (same values, same comparisons, same swaps)
  std::vector<std::unique_ptr<ExistingNode>> existing_store;
  std::vector<ExistingNode*> existing_nodes;
  std::map<std::string, ResourceList> remaining_resources;  // nodepool -> remaining (only pools present in the map count)
  std::unordered_map<Sym, PodData> cached;                  // by uid
  Topology topology;
  ReservationManager reservations;
  bool tolerate_prefer_no_schedule = false;
  long long node_id = 0;  // hostname-placeholder counter (nodeclaim.go:83,93); per-solve in the oracle
  Counters ctr;
  int last_err = 0, last_diag = 0;
  struct EvalCtx { Counters ctr; int err = 0, diag = 0; };
  // parallelizeUntil (scheduler.go:939-961): the reference evaluates the candidates of addToInflightNode on a worker pool and
  // keeps the lowest index that succeeds (:667-686). `threads` > 1 does the same here — same winner, same counters as the
  // sequential scan (every claim up to the winner is counted once); used for the offline pins of the largest configurations.
  int threads = 1;
  size_t parallel_min = 512;          // scans shorter than this stay sequential
  struct Pool;
  std::shared_ptr<Pool> pool;
  long long progress_every = 0;   // ORACLE_PROGRESS: a line on stderr every so many pods (the hour-long offline pins)
  double t_sort = 0, t_scan = 0;   // seconds in sort.Slice / in the in-flight scan (ORACLE_TIMING prints them)
  struct ParFound { size_t j = 0; Requirements r; std::vector<const InstanceType*> its; std::vector<const Offering*> ofs; };
  std::vector<ParFound> par_found;   // per worker: the lowest candidate it found to pass, with CanAdd's outputs
  std::vector<unsigned long long> par_it_evals;   // per-candidate instance-type evaluation counts of the scan in flight

  // ---- pod requirement derivation: requirements.go:74-118 --------------------------------------------------
  static Requirements pod_requirements(Pod& p, bool required_only) {
    Requirements r = label_requirements(p.node_selector);
    if (!p.has_node_affinity) return r;
    if (!required_only && !p.preferred_terms.empty()) {
      // sort.Slice(preferred, weight desc) — mutates the pod's slice in place (requirements.go:102)
      go_sort_slice(p.preferred_terms, [](const PreferredSchedulingTerm& a, const PreferredSchedulingTerm& b) { return a.weight > b.weight; });
      r.add_all(exprs_to_requirements(p.preferred_terms[0].preference));
    }
    if (p.has_required && !p.required_terms.empty()) r.add_all(exprs_to_requirements(p.required_terms[0]));
    return r;
  }
  // updateCachedPodData — scheduler.go:554-580
  void update_cached_pod_data(Pod& p) {
    PodData d;
    d.reqs = pod_requirements(p, opts.ignore_preferences);
    d.strict_reqs = d.reqs;
    if (p.has_node_affinity && !p.preferred_terms.empty()) d.strict_reqs = pod_requirements(p, true);
    d.requests = p.requests;                       // RequestsForPods — resources.go:30-38
    d.requests[W().pods] = (i128)1 * 1000000000;
    for (auto& alt : p.volume_requirements) d.volume_reqs.push_back(exprs_to_requirements(alt));   // scheduler.go:572
    cached[p.uid_s] = d;
  }

  // ---- Preferences — preferences.go:38-146 -----------------------------------------------------------------
  static bool remove_required_node_affinity_term(Pod& p) {
    if (!p.has_node_affinity || !p.has_required || p.required_terms.empty()) return false;
    if (p.required_terms.size() > 1) { p.required_terms.erase(p.required_terms.begin()); return true; }
    return false;
  }
  template <class T> static void stable_by_weight_desc(std::vector<T>& v) {
    std::stable_sort(v.begin(), v.end(), [](const T& a, const T& b) { return a.weight > b.weight; });  // sort.SliceStable
  }
  static bool remove_preferred_pod_affinity_term(Pod& p) {
    if (!p.has_pod_affinity || p.affinity_preferred.empty()) return false;
    stable_by_weight_desc(p.affinity_preferred);
    p.affinity_preferred.erase(p.affinity_preferred.begin());
    return true;
  }
  static bool remove_preferred_pod_anti_affinity_term(Pod& p) {
    if (!p.has_pod_anti_affinity || p.anti_preferred.empty()) return false;
    stable_by_weight_desc(p.anti_preferred);
    p.anti_preferred.erase(p.anti_preferred.begin());
    return true;
  }
  static bool remove_preferred_node_affinity_term(Pod& p) {
    if (!p.has_node_affinity || p.preferred_terms.empty()) return false;
    stable_by_weight_desc(p.preferred_terms);
    p.preferred_terms.erase(p.preferred_terms.begin());
    return true;
  }
  static bool remove_topology_spread_schedule_anyway(Pod& p) {
    for (size_t i = 0; i < p.tscs.size(); ++i)
      if (p.tscs[i].when_unsatisfiable == W().ScheduleAnyway) {
        p.tscs[i] = p.tscs.back();
        p.tscs.pop_back();
        return true;
      }
    return false;
  }
  static bool tolerate_prefer_no_schedule_taints(Pod& p) {
    // MatchToleration: key, operator, value, effect all equal
    const WellKnownSyms& w = W();
    for (auto& t : p.tolerations) if (t.key == w.empty && t.op == w.Exists && t.value == w.empty && t.effect == w.PreferNoSchedule) return false;
    p.tolerations.push_back({w.empty, w.Exists, w.empty, w.PreferNoSchedule});
    return true;
  }
  bool relax(Pod& p) {
    if (remove_required_node_affinity_term(p)) return true;
    if (remove_preferred_pod_affinity_term(p)) return true;
    if (remove_preferred_pod_anti_affinity_term(p)) return true;
    if (remove_preferred_node_affinity_term(p)) return true;
    if (remove_topology_spread_schedule_anyway(p)) return true;
    if (tolerate_prefer_no_schedule && tolerate_prefer_no_schedule_taints(p)) return true;
    return false;
  }

  // ---- daemon overhead: scheduler.go:972-1043 --------------------------------------------------------------
  bool daemon_pod_compatible(const NodeClaimTemplate& nct, const InstanceType& it, const Pod& pod_in) {
    Pod p = pod_in;
    tolerate_prefer_no_schedule_taints(p);
    if (!taints_tolerated(nct.taints, p.tolerations)) return false;
    for (;;) {
      Requirements pr_ = pod_requirements(p, true);
      if (nct.reqs.compatible(pr_, true) && it.reqs.intersects(pr_)) return true;
      if (!remove_required_node_affinity_term(p)) return false;
    }
  }
  void build_daemon_overhead_groups(NodeClaimTemplate& nct) {
    // groups keyed by the sorted set of compatible daemon pods (podSetKey). lo.Values(groups) randomises group order in
    // the reference (scheduler.go:1001); canonicalised here to order of first appearance.
    std::vector<std::pair<std::string, DaemonOverheadGroup>> groups;
    for (auto* it : nct.its) {
      std::vector<const Pod*> compat;
      for (auto& dp : pr->daemonset_pods) if (daemon_pod_compatible(nct, *it, dp)) compat.push_back(&dp);
      std::vector<std::string> keys;
      for (auto* p : compat) keys.push_back(str(p->ns) + "/" + p->name);
      std::sort(keys.begin(), keys.end());
      std::string key;
      for (auto& k : keys) { if (!key.empty()) key += ","; key += k; }
      DaemonOverheadGroup* g = nullptr;
      for (auto& e : groups) if (e.first == key) { g = &e.second; break; }
      if (!g) {
        DaemonOverheadGroup ng;
        if (!compat.empty()) {
          for (auto* p : compat) ng.overhead = res_merge(ng.overhead, p->requests);
          ng.overhead[W().pods] = (i128)compat.size() * 1000000000;
          for (auto* p : compat) ng.host_ports.insert(ng.host_ports.end(), p->host_ports.begin(), p->host_ports.end());
        }
        groups.push_back({key, ng});
        g = &groups.back().second;
      }
      g->its.insert(it);
      g->its_ordered.push_back(it);
    }
    for (auto& e : groups) nct.daemon_groups.push_back(e.second);
  }

  // ---- NewScheduler — scheduler.go:127-215 (+ Provisioner.NewScheduler ordering, provisioner.go:293) --------
  // `probe_pods` / `removed`: oracle_sweep_json simulates many candidate sets of one cluster document (possibly on several
  // threads): the pods of this simulation and the state nodes (by index) that are not part of it (helpers.go:76-80)
  const std::vector<char>* removed = nullptr;
  bool is_removed(const StateNode& n) const { return removed && (*removed)[(size_t)(&n - pr->state_nodes.data())]; }
  void init(const Problem& problem, const std::vector<Pod>* probe_pods = nullptr, const std::vector<char>* removed_nodes = nullptr) {
    pr = &problem;
    opts = problem.opts;
    removed = removed_nodes;
    pods = probe_pods ? *probe_pods : problem.pods;
    for (auto& np : problem.node_pools) if (!np.is_static) pools.push_back(&np);
    // OrderByWeight — pkg/utils/nodepool/nodepool.go:161-171 (total order: weight desc, name desc)
    std::sort(pools.begin(), pools.end(), [](const NodePool* a, const NodePool* b) { return a->weight != b->weight ? a->weight > b->weight : a->name > b->name; });
    for (auto* np : pools) for (auto& t : np->taints) if (t.effect == W().PreferNoSchedule) tolerate_prefer_no_schedule = true;

    std::vector<const StateNode*> snodes;
    for (auto& n : problem.state_nodes) if (!is_removed(n)) snodes.push_back(&n);
    topology.init(problem, pools, snodes, pods, opts.ignore_preferences);

    for (auto* np : pools) {
      if (np->instance_types.empty()) continue;  // provisioner.go:311-314 skips pools with no resolved instance types
      // NewNodeClaimTemplate — nodeclaimtemplate.go:66-94
      NodeClaimTemplate nct;
      nct.np = np; nct.nodepool_name = np->name; nct.weight = np->weight; nct.taints = np->taints;
      nct.reqs.add_all(exprs_to_requirements(np->requirements));
      SymMap labels = np->labels;
      labels.set(W().nodepool, sym(np->name));
      labels.set(np->node_class_label_key, np->node_class_name);
      nct.reqs.add_all(label_requirements(labels));
      nct.reqs.add(Requirement::make(W().registered, Op::In, W().true_));
      nct.reqs.add(Requirement::make(W().initialized, Op::In, W().true_));
      // prefilter — scheduler.go:159
      std::vector<const InstanceType*> all;
      for (int idx : np->instance_types) all.push_back(&problem.catalog[idx]);
      DaemonOverheadGroup g0;
      g0.its.insert(all.begin(), all.end());
      g0.its_ordered = all;
      std::map<Sym, int> unsat;
      FilterDiag d;
      ResourceList none;
      std::vector<const InstanceType*> remaining;
      filter_instance_types(all, nct.reqs, {g0}, none, opts.min_values_best_effort, remaining, unsat, d, nullptr);
      if (remaining.empty()) continue;
      nct.its = remaining;
      templates.push_back(nct);
    }
    for (size_t i = 0; i < templates.size(); ++i) { templates[i].index = (int)i; build_daemon_overhead_groups(templates[i]); }
    for (auto* np : pools) if (np->has_limits) remaining_resources[np->name] = np->limits;
    // NOTE scheduler.go:183 builds the map for every NodePool (nil limits => empty ResourceList, which never filters);
    // only pools with limits can change behaviour, so only those are kept.
    reservations.init(problem, pools);
    // calculateExistingNodeClaims — scheduler.go:792-802
    for (auto& n : problem.state_nodes) {
      if (is_removed(n)) continue;
      auto en = std::make_unique<ExistingNode>();
      en->node = &n;
      en->host_ports = n.host_ports;
      en->volumes = n.volumes;
      // daemons compatible with the node (scheduler.go:805-832) minus what already runs there (existingnode.go:50-60)
      ResourceList daemon;
      int ndaemons = 0;
      for (auto& dp : problem.daemonset_pods) {
        Pod p = dp;
        if (!taints_tolerated(n.taints, p.tolerations)) continue;
        if (!label_requirements(n.labels).compatible(pod_requirements(p, true), false)) continue;
        daemon = res_merge(daemon, p.requests);
        ndaemons++;
      }
      daemon[W().pods] = (i128)ndaemons * 1000000000;
      res_subtract_from(daemon, n.daemonset_requests);
      for (auto& kv : daemon) if (kv.second < 0) kv.second = 0;
      en->remaining = res_subtract(n.available, daemon);
      en->reqs = label_requirements(n.labels);
      en->reqs.add(Requirement::make(W().hostname, Op::In, n.hostname));
      en->under_consolidate_after = opts.enforce_consolidate_after && n.under_consolidate_after;
      topology.reg(W().hostname, n.hostname);
      // updateRemainingResources — scheduler.go:835-842
      auto* npit = n.labels.find(W().nodepool);
      if (npit) {
        auto rr = remaining_resources.find(str(npit->second));
        if (rr != remaining_resources.end()) rr->second = res_subtract(rr->second, n.capacity);
      }
      existing_nodes.push_back(en.get());
      existing_store.push_back(std::move(en));
    }
    // sortExistingNodes — scheduler.go:845-858 (SliceStable; total order)
    std::stable_sort(existing_nodes.begin(), existing_nodes.end(), [](const ExistingNode* a, const ExistingNode* b) {
      if (a->node->initialized != b->node->initialized) return a->node->initialized;
      return a->node->name < b->node->name;
    });
  }

  // ---- ExistingNode.CanAdd / Add — existingnode.go:81-185 --------------------------------------------------
  bool existing_can_add(ExistingNode& n, const Pod& pod, const PodData& pd, Requirements& out) {
    ctr.bin_evaluations++;
    if (!taints_tolerated(n.node->taints, pod.tolerations)) { last_err = ERR_TAINTS; return false; }
    // VolumeUsage.ExceedsLimits (volumeusage.go:193-200; existingnode.go:88): over the UNION of the node's and the pod's volumes,
    // every driver with a limit — also one the pod adds nothing to
    for (auto& kv : n.volumes) {
      auto lim = n.node->volume_limits.find(kv.first);
      if (lim == n.node->volume_limits.end()) continue;
      SymSet u = kv.second;
      auto pv = pod.volumes.find(kv.first);
      if (pv != pod.volumes.end()) u.insert(pv->second.begin(), pv->second.end());
      if ((int)u.size() > lim->second) { last_err = ERR_EXISTING; return false; }
    }
    for (auto& kv : pod.volumes) {
      if (n.volumes.count(kv.first)) continue;
      auto lim = n.node->volume_limits.find(kv.first);
      if (lim != n.node->volume_limits.end() && (int)kv.second.size() > lim->second) { last_err = ERR_EXISTING; return false; }
    }
    if (host_ports_conflict(pod.host_ports, n.host_ports)) { last_err = ERR_EXISTING; return false; }   // existingnode.go:87-93
    if (!res_fits(pd.requests, n.remaining)) { last_err = ERR_RESOURCES; return false; }
    if (!n.reqs.compatible(pd.reqs, false)) { last_err = ERR_INCOMPATIBLE; return false; }
    // the node's requirements + the pod's. Copies are made when something is written: adding an empty set leaves the set as it
    // is, and a candidate that fails before its requirements are returned needs none (the reference's copies are Go values)
    Requirements merged;
    const Requirements* base0 = &n.reqs;
    if (!pd.reqs.m.empty()) { merged = n.reqs; merged.add_all(pd.reqs); base0 = &merged; }
    // volume requirement alternatives — existingnode.go:108-139; tryVolumeAlternative :143-168. They narrow the node's
    // requirements only: topology counts with the pod's own (strict) requirements.
    const size_t n_alt = pd.volume_reqs.empty() ? 1 : pd.volume_reqs.size();
    Requirements alt_base;
    for (size_t a = 0; a < n_alt; ++a) {
      const Requirements* base = base0;
      if (!pd.volume_reqs.empty()) {
        if (!base0->compatible(pd.volume_reqs[a], false)) { last_err = ERR_INCOMPATIBLE; continue; }
        alt_base = *base0;
        alt_base.add_all(pd.volume_reqs[a]);
        base = &alt_base;
      }
      Requirements topo;
      if (!topology.add_requirements(pod, n.node->taints, pd.strict_reqs, *base, topo)) { last_err = ERR_TOPOLOGY; continue; }
      if (!base->compatible(topo, false)) { last_err = ERR_TOPOLOGY; continue; }
      out = *base;
      out.add_all(topo);
      return true;
    }
    return false;
  }
  void existing_add(ExistingNode& n, Pod* pod, const PodData& pd, const Requirements& reqs) {
    n.pods.push_back(pod);
    res_subtract_from(n.remaining, pd.requests);
    n.reqs = reqs;
    n.host_ports.insert(n.host_ports.end(), pod->host_ports.begin(), pod->host_ports.end());   // existingnode.go:178
    for (auto& kv : pod->volumes) n.volumes[kv.first].insert(kv.second.begin(), kv.second.end());  // VolumeUsage.Add — existingnode.go:179, volumeusage.go:206-209
    topology.record(*pod, n.node->taints, reqs);
  }

  // ---- NodeClaim — nodeclaim.go:85-350 ---------------------------------------------------------------------
  std::unique_ptr<NodeClaim> new_node_claim(const NodeClaimTemplate& t, const std::vector<const InstanceType*>& its) {
    auto nc = std::make_unique<NodeClaim>();
    char buf[64];
    snprintf(buf, sizeof buf, "hostname-placeholder-%04lld", ++node_id);
    nc->hostname = sym(buf);
    nc->tmpl = &t;
    nc->reqs = t.reqs;
    nc->reqs.add(Requirement::make(W().hostname, Op::In, nc->hostname));
    nc->its = its;
    return nc;
  }
  // offeringsToReserve — nodeclaim.go:303-350
  bool offerings_to_reserve(NodeClaim& n, const std::vector<const InstanceType*>& its, const Requirements& reqs, std::vector<const Offering*>& out) {
    out.clear();
    if (!opts.reserved_capacity) return true;
    bool has_compatible = false;
    for (auto* it : its)
      for (auto& o : it->offerings) {
        if (o.capacity_type() != W().reserved || !o.available) continue;
        if (!reqs.compatible(o.reqs, true)) continue;
        has_compatible = true;
        if (reservations.can_reserve(n.hostname, o)) out.push_back(&o);
      }
    if (opts.reserved_offering_strict) {
      if (has_compatible && out.empty()) return false;
      if (!n.reserved_offerings.empty() && out.empty()) return false;
    }
    return true;
  }
  // CanAdd — nodeclaim.go:124-242
  bool claim_can_add(NodeClaim& n, const Pod& pod, const PodData& pd, bool relax_min_values, Requirements& out_reqs,
                     std::vector<const InstanceType*>& out_its, std::vector<const Offering*>& out_ofs, EvalCtx* cx = nullptr) {
    // cx: the counters and error slots of a worker of the parallel in-flight scan (add_to_inflight); none: the scheduler's own
    Counters& ctr = cx ? cx->ctr : this->ctr;
    int& last_err = cx ? cx->err : this->last_err;
    int& last_diag = cx ? cx->diag : this->last_diag;
    ctr.bin_evaluations++;
    last_diag = 0;
    if (!taints_tolerated(n.tmpl->taints, pod.tolerations)) { last_err = ERR_TAINTS; return false; }
    // nodeclaim.go:130-136: the reference copies the claim's requirements, then tests Compatible on the copy, then Adds;
    // Compatible does not write, so the copy is made once the test has passed
    if (!n.reqs.compatible(pd.reqs, true)) { last_err = ERR_INCOMPATIBLE; return false; }
    // (the same goes for the Add: a pod without requirements of its own adds nothing, and a candidate that fails topology
    // never hands its set on — the working copy is made on the way to the instance-type filter)
    Requirements merged;
    const Requirements* base0 = &n.reqs;
    if (!pd.reqs.m.empty()) { merged = n.reqs; merged.add_all(pd.reqs); base0 = &merged; }
    // volume requirement alternatives — nodeclaim.go:138-157; tryVolumeAlternative :164-242: the first alternative that passes
    // topology, the instance-type filter and the reservation check wins; the error reported is the last alternative's
    const size_t n_alt = pd.volume_reqs.empty() ? 1 : pd.volume_reqs.size();
    Requirements alt_base;
    for (size_t a = 0; a < n_alt; ++a) {
      // every alternative starts from the claim's requirements + the pod's
      const Requirements* base_in = base0;
      last_diag = 0;
      if (!pd.volume_reqs.empty()) {
        if (!base0->compatible(pd.volume_reqs[a], true)) { last_err = ERR_INCOMPATIBLE; continue; }
        alt_base = *base0;
        alt_base.add_all(pd.volume_reqs[a]);
        base_in = &alt_base;
      }
      Requirements topo;
      if (!topology.add_requirements(pod, n.tmpl->taints, pd.strict_reqs, *base_in, topo)) { last_err = ERR_TOPOLOGY; continue; }
      if (!base_in->compatible(topo, true)) { last_err = ERR_TOPOLOGY; continue; }
      Requirements base = *base_in;
      base.add_all(topo);
      ResourceList requests = res_merge(n.requests, pd.requests);
      std::map<Sym, int> unsat;
      FilterDiag d;
      std::vector<const InstanceType*> its;
      bool ok = filter_instance_types(n.its, base, n.tmpl->daemon_groups, requests, relax_min_values, its, unsat, d, &ctr, &pod.host_ports, &n.host_ports);
      if (relax_min_values) for (auto& kv : unsat) {
        Requirement* r = base.find(kv.first);   // a key of `base` by construction (SatisfiesMinValues walks its keys)
        if (!r) throw std::runtime_error("minValues relaxation names a key the requirements do not have");
        r->min_values = kv.second;
      }
      if (!ok) { last_err = d.min_values_incompatible ? ERR_MIN_VALUES : ERR_INSTANCE_TYPES; last_diag = d.bits(); continue; }
      std::vector<const Offering*> ofs;
      if (!offerings_to_reserve(n, its, base, ofs)) { last_err = ERR_RESERVED; continue; }
      out_reqs = base; out_its = its; out_ofs = ofs;
      return true;
    }
    return false;
  }
  // Add — nodeclaim.go:247-263
  void claim_add(NodeClaim& n, Pod* pod, const PodData& pd, const Requirements& reqs, const std::vector<const InstanceType*>& its, const std::vector<const Offering*>& ofs) {
    n.pods.push_back(pod);
    n.its = its;
    n.requests = res_merge(n.requests, pd.requests);
    n.reqs = reqs;
    n.host_ports.insert(n.host_ports.end(), pod->host_ports.begin(), pod->host_ports.end());
    topology.reg(W().hostname, n.hostname);
    topology.record(*pod, n.tmpl->taints, reqs);
    reservations.reserve(n.hostname, ofs);
    SymSet updated;
    for (auto* o : ofs) updated.insert(o->reservation_id());
    for (auto* o : n.reserved_offerings) if (!updated.count(o->reservation_id())) reservations.release(n.hostname, *o);
    n.reserved_offerings = ofs;
  }

  // ---- add — scheduler.go:582-790 --------------------------------------------------------------------------
  // subtractMax — scheduler.go:1049-1066 ; filterByRemainingResources — :1069-1085
  static ResourceList subtract_max(const ResourceList& remaining, const std::vector<const InstanceType*>& its) {
    if (its.empty()) return remaining;
    std::vector<const ResourceList*> caps;
    for (auto* it : its) caps.push_back(&it->capacity);
    ResourceList mx = res_max(caps), out;
    for (auto& kv : remaining) { auto f = mx.find(kv.first); out[kv.first] = kv.second - (f == mx.end() ? 0 : f->second); }
    return out;
  }
  static std::vector<const InstanceType*> filter_by_remaining(const std::vector<const InstanceType*>& its, const ResourceList& remaining) {
    std::vector<const InstanceType*> out;
    for (auto* it : its) {
      bool viable = true;
      for (auto& kv : remaining) { auto f = it->capacity.find(kv.first); i128 c = f == it->capacity.end() ? 0 : f->second; if (c > kv.second) viable = false; }
      if (viable) out.push_back(it);
    }
    return out;
  }
  bool add_to_new_claim(Pod& pod, Pod* queue_pod) {
    const PodData& pd = cached[pod.uid_s];
    int first_err = 0, first_diag = 0;
    for (auto& t : templates) {
      std::vector<const InstanceType*> its = t.its;
      auto rr = remaining_resources.find(t.nodepool_name);
      if (rr != remaining_resources.end()) {
        auto nodes = rr->second.find(W().nodes);
        if (nodes != rr->second.end() && nodes->second == 0) { if (!first_err) first_err = ERR_LIMITS; continue; }
        its = filter_by_remaining(its, rr->second);
        if (its.empty()) { if (!first_err) first_err = ERR_LIMITS; continue; }
      }
      std::unique_ptr<NodeClaim> nc = new_node_claim(t, its);
      Requirements r; std::vector<const InstanceType*> rem; std::vector<const Offering*> ofs;
      if (!claim_can_add(*nc, pod, pd, opts.min_values_best_effort, r, rem, ofs)) {
        if (!first_err) { first_err = last_err; first_diag = last_diag; }
        if (last_err == ERR_RESERVED) { last_err = ERR_RESERVED; last_diag = 0; return false; }  // :736-751: voids later templates
        continue;
      }
      // minValuesRelaxed annotation — scheduler.go:763-772
      bool relaxed = false;
      for (auto& q : nc->reqs.m) {
        auto upd = r.get(q.key).min_values;
        auto orig = q.min_values;
        if (orig && upd && *upd < *orig) relaxed = true;
      }
      nc->annotations[kMinValuesRelaxedAnnotation] = relaxed ? "true" : "false";
      claim_add(*nc, queue_pod, pd, r, rem, ofs);
      nc->id = (int)claim_store.size();
      new_node_claims.push_back(nc.get());
      claim_pods.push_back((uint32_t)nc->pods.size());
      if (rr != remaining_resources.end()) rr->second = subtract_max(rr->second, nc->its);
      claim_store.push_back(std::move(nc));
      return true;
    }
    last_err = first_err ? first_err : ERR_NO_TEMPLATES;
    last_diag = first_diag;
    return false;
  }
  bool add(Pod& pod, Pod* queue_pod) {
    if (add_to_existing(pod, queue_pod)) return true;
    // sort.Slice(s.newNodeClaims, len(Pods) asc) — scheduler.go:598, Go pdqsort (unstable)
    ctr.sorts++;
    const auto tq0 = std::chrono::steady_clock::now();
    {
      auto less = [&](int i, int j) { return claim_pods[(size_t)i] < claim_pods[(size_t)j]; };
      auto swap = [&](int i, int j) { std::swap(new_node_claims[(size_t)i], new_node_claims[(size_t)j]); std::swap(claim_pods[(size_t)i], claim_pods[(size_t)j]); };
      GoSort<decltype(less), decltype(swap)> srt(less, swap);
      srt.sort_slice((int)new_node_claims.size());
    }
    const auto tq1 = std::chrono::steady_clock::now();
    const bool placed = add_to_inflight(pod, queue_pod);
    const auto tq2 = std::chrono::steady_clock::now();
    t_sort += std::chrono::duration<double>(tq1 - tq0).count(); t_scan += std::chrono::duration<double>(tq2 - tq1).count();
    if (placed) return true;
    if (templates.empty()) { last_err = ERR_NO_TEMPLATES; last_diag = 0; return false; }
    return add_to_new_claim(pod, queue_pod);
  }
  // The reference appends the *relaxed copy* it was handed to Pods; results are reported by uid so the distinction is
  // invisible. queue_pod keeps the pointer stable.
  bool add_to_existing(Pod& pod, Pod* queue_pod) {
    const PodData& pd = cached[pod.uid_s];
    for (auto* en : existing_nodes) {
      if (en->under_consolidate_after && (pod.phase != W().Pending && !pr->deleting_node_names.count(pod.node_name))) continue;
      Requirements r;
      if (existing_can_add(*en, pod, pd, r)) { existing_add(*en, queue_pod, pd, r); return true; }
    }
    return false;
  }
  bool add_to_inflight(Pod& pod, Pod* queue_pod) {
    const PodData& pd = cached[pod.uid_s];
    if (threads > 1 && new_node_claims.size() >= parallel_min) return add_to_inflight_parallel(pod, queue_pod, pd);
    const size_t n_claims = new_node_claims.size();
    for (size_t i = 0; i < n_claims; ++i) {
      NodeClaim* nc = new_node_claims[i];
      if (i + 4 < n_claims) { const NodeClaim* ahead = new_node_claims[i + 4]; __builtin_prefetch(ahead); __builtin_prefetch(ahead->reqs.m.data()); }   // memory latency only
      Requirements r; std::vector<const InstanceType*> its; std::vector<const Offering*> ofs;
      if (claim_can_add(*nc, pod, pd, false, r, its, ofs)) { claim_add(*nc, queue_pod, pd, r, its, ofs); claim_pods[i] = (uint32_t)nc->pods.size(); return true; }
    }
    return false;
  }

  bool add_to_inflight_parallel(Pod& pod, Pod* queue_pod, const PodData& pd);

  // trySchedule — scheduler.go:521-552 (p is the DeepCopy)
  bool try_schedule(Pod& copy, Pod* queue_pod) {
    for (;;) {
      if (add(copy, queue_pod)) return true;
      if (last_err == ERR_RESERVED) return false;
      int err = last_err, diag = last_diag;
      if (!relax(copy)) { last_err = err; last_diag = diag; return false; }
      ctr.relaxations++;
      topology.update(copy);
      update_cached_pod_data(copy);
    }
  }

  // Solve — scheduler.go:440-519 ; Queue — queue.go:31-108
  Results solve() {
    Results res;
    for (auto& p : pods) update_cached_pod_data(p);
    std::vector<Pod*> q;
    for (auto& p : pods) q.push_back(&p);
    // NewQueue: sort.Slice(byCPUAndMemoryDescending) — total order, any sort gives the same result (queue.go:37-41,72-108)
    std::sort(q.begin(), q.end(), [&](Pod* a, Pod* b) {
      const ResourceList& l = cached[a->uid_s].requests; const ResourceList& r = cached[b->uid_s].requests;
      auto get = [](const ResourceList& m, Sym k) { auto it = m.find(k); return it == m.end() ? (i128)0 : it->second; };
      i128 lc = get(l, W().cpu), rc = get(r, W().cpu);
      if (lc != rc) return lc > rc;
      i128 lm = get(l, W().memory), rm = get(r, W().memory);
      if (lm != rm) return lm > rm;
      if (a->creation != b->creation) return a->creation < b->creation;
      return a->uid < b->uid;
    });
    std::unordered_map<Sym, size_t> last_len;
    size_t head = 0;
    long long steps = 0;
    for (;;) {
      size_t qlen = q.size() - head;
      if (qlen == 0) break;
      Pod* p = q[head];
      auto ll = last_len.find(p->uid_s);
      if (ll != last_len.end() && ll->second == qlen) break;   // queue.go:52-56 (checked before popping)
      if (opts.max_steps >= 0 && steps >= opts.max_steps) { res.timed_out = true; break; }  // ctx deadline stand-in
      head++;
      steps++;
      ctr.pops++;
      if (progress_every > 0 && steps % progress_every == 0) fprintf(stderr, "oracle progress: %lld pods popped, %zu NodeClaims, %.0f s in the in-flight scan, %.0f s in sort.Slice\n", steps, new_node_claims.size(), t_scan, t_sort);
      Pod copy = *p;  // pod.DeepCopy()
      if (try_schedule(copy, p)) {
        res.pod_errors.erase(p->uid);
      } else {
        res.pod_errors[p->uid] = {last_err, last_diag};
        topology.update(*p);
        update_cached_pod_data(*p);
        q.push_back(p);
        last_len[p->uid_s] = q.size() - head;   // queue.go:63-66
      }
    }
    // FinalizeScheduling — nodeclaim.go:383-409
    for (auto* nc : new_node_claims) {
      nc->reqs.erase(W().hostname);
      if (!nc->reserved_offerings.empty()) {
        nc->reqs.put(Requirement::make(W().capacity_type, Op::In, W().reserved));
        std::vector<Sym> ids;
        for (auto* o : nc->reserved_offerings) ids.push_back(o->reservation_id());
        nc->reqs.add(Requirement::make(W().reservation_id, Op::In, std::nullopt, ids));
      }
      // addDaemonRequests — nodeclaim.go:353-377
      std::set<const InstanceType*> remaining(nc->its.begin(), nc->its.end());
      ResourceList min_overhead;
      bool have = false;
      for (auto& g : nc->tmpl->daemon_groups) {
        bool has_remaining = false;
        for (auto* it : g.its_ordered) if (remaining.count(it)) { has_remaining = true; break; }
        if (!has_remaining) continue;
        if (min_overhead.empty()) min_overhead = g.overhead;
        else min_overhead = res_min({&min_overhead, &g.overhead});
        have = true;
      }
      (void)have;
      if (!min_overhead.empty()) nc->requests = res_merge(nc->requests, min_overhead);
    }
    res.new_node_claims = new_node_claims;
    res.existing_nodes = existing_nodes;
    return res;
  }
};


// A persistent worker pool for the parallel in-flight scan: one job at a time, the caller takes part. The workers spin on the
// generation counter between jobs (a job arrives with every pod; a condition variable's wake-up costs more than a scan).
struct Scheduler::Pool {
  std::vector<std::thread> workers;
  const std::function<void(int)>* job = nullptr;
  std::atomic<long long> generation{0};
  std::atomic<int> pending{0};
  std::atomic<bool> stop{false};
  explicit Pool(int n) {
    for (int t = 1; t < n; ++t) workers.emplace_back([this, t]() {
      long long seen = 0;
      for (;;) {
        int spins = 0;
        while (generation.load(std::memory_order_acquire) == seen) {
          if (stop.load(std::memory_order_relaxed)) return;
          __builtin_ia32_pause();   // (a spinning sibling hyperthread without it takes issue slots from the one that works)
          if (++spins > 20000) { std::this_thread::yield(); spins = 0; }
        }
        seen = generation.load(std::memory_order_acquire);
        (*job)(t);
        pending.fetch_sub(1, std::memory_order_acq_rel);
      }
    });
  }
  ~Pool() { stop = true; for (auto& w : workers) w.join(); }
  void run(const std::function<void(int)>& f) {
    job = &f;
    pending.store((int)workers.size(), std::memory_order_release);
    generation.fetch_add(1, std::memory_order_acq_rel);
    f(0);
    while (pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
  }
};

inline bool Scheduler::add_to_inflight_parallel(Pod& pod, Pod* queue_pod, const PodData& pd) {
  if (!pool) pool = std::make_shared<Pool>(threads);
  const size_t n = new_node_claims.size();
  std::atomic<size_t> best(n);
  // (every claim below the winner is evaluated exactly once, so the entries read below are all written: no zero-fill per pod)
  if (par_it_evals.size() < n) par_it_evals.resize(n + n / 2 + 64);
  unsigned long long* const it_evals = par_it_evals.data();
  // chunks of eight candidates dealt round-robin (worker t takes chunks t, t + T, ...): no shared cursor to fight over, and a worker
  // keeps meeting the same claims from pod to pod. A worker stops at the first chunk that starts beyond the best index so far.
  const size_t chunk = 8, T = (size_t)threads;
  // a worker keeps the outputs of the lowest candidate IT found to pass: the winner's are among them, no second evaluation
  std::vector<ParFound>& found = par_found;   // (kept across pods: no T constructions and destructions per pod)
  if (found.size() != T) found.resize(T);
  for (auto& f : found) f.j = n;
  pool->run([&](int t) {
    ParFound& mine = found[(size_t)t];
    for (size_t i0 = (size_t)t * chunk; i0 < n; i0 += T * chunk) {
      if (i0 > best.load(std::memory_order_relaxed)) return;
      for (size_t j = i0; j < std::min(n, i0 + chunk); ++j) {
        if (j > best.load(std::memory_order_relaxed)) return;
        EvalCtx cx;
        Requirements r; std::vector<const InstanceType*> its; std::vector<const Offering*> ofs;
        const bool ok = claim_can_add(*new_node_claims[j], pod, pd, false, r, its, ofs, &cx);
        it_evals[j] = (unsigned long long)cx.ctr.it_evaluations;
        if (ok) {
          if (j < mine.j) { mine.j = j; mine.r = std::move(r); mine.its = std::move(its); mine.ofs = std::move(ofs); }
          size_t cur = best.load(); while (j < cur && !best.compare_exchange_weak(cur, j)) {}
          return;   // (everything this worker would still look at lies beyond j)
        }
      }
    }
  });
  const size_t w = best.load();
  // the sequential scan's counters: every claim up to and including the winner once
  const size_t counted = w < n ? w + 1 : n;
  ctr.bin_evaluations += (long long)counted;
  for (size_t j = 0; j < counted; ++j) ctr.it_evaluations += (long long)it_evals[j];
  if (w == n) return false;
  ParFound* win = nullptr;
  for (auto& f : found) if (f.j == w) win = &f;
  if (!win) throw std::runtime_error("parallel in-flight scan: the winner's outputs are missing");
  claim_add(*new_node_claims[w], queue_pod, pd, win->r, win->its, win->ofs);
  claim_pods[w] = (uint32_t)new_node_claims[w]->pods.size();
  return true;
}

}  // namespace oracle
