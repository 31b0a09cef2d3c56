// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Problem model (the inputs Provisioner.NewScheduler assembles, pkg/controllers/provisioning/provisioner.go:265-360)
// and the cloudprovider value types (pkg/cloudprovider/types.go:123-142, 470-486), parsed from the problem JSON.
#pragma once
#include <cmath>
#include <cfloat>

#include "algebra.hpp"
#include "json_mini.hpp"
#include "pdqsort.hpp"

namespace oracle {

struct NodeSelectorExpr {
  Sym key;
  Op op;
  std::vector<Sym> values;
  std::optional<int> min_values;
};
typedef std::vector<NodeSelectorExpr> NodeSelectorTerm;
struct PreferredSchedulingTerm { int weight = 0; NodeSelectorTerm preference; };
struct PodAffinityTerm {
  LabelSelector selector;
  Sym topology_key = kNoSym;
  std::vector<Sym> namespaces;
  LabelSelector namespace_selector;   // nil = absent
};
struct WeightedPodAffinityTerm { int weight = 0; PodAffinityTerm term; };
struct TopologySpreadConstraint {
  int max_skew = 1;
  Sym topology_key = kNoSym;
  Sym when_unsatisfiable = W().DoNotSchedule;
  LabelSelector selector;
  std::optional<int> min_domains;
  std::optional<Sym> node_taints_policy, node_affinity_policy;
  std::vector<Sym> match_label_keys;
};

// corev1.Pod subset read by the path. `requests` is the output of resourcehelper.PodRequests (flattened upstream
// of the boundary); RequestsForPods adds pods=1 (resources.go:36).
// scheduling.HostPort — hostportusage.go:39-62. The IP is kept as the canonical text net.ParseIP would print ("" and
// unparsable text become one nil value, IPv4-mapped IPv6 folds to IPv4); Matches: same protocol and port, and equal IPs or
// one of them unspecified (0.0.0.0 or ::).
struct HostPort {
  Sym ip = kNoSym;
  int port = 0;
  Sym protocol = kNoSym;
  bool unspecified() const { return ip == W().ip_any4 || ip == W().ip_any6; }
  bool matches(const HostPort& o) const {
    if (protocol != o.protocol || port != o.port) return false;
    return ip == o.ip || unspecified() || o.unspecified();
  }
};
inline std::string canonical_ip(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  if (s.rfind("::ffff:", 0) == 0 && s.find('.') != std::string::npos) s = s.substr(7);
  if (s.find(':') != std::string::npos) {   // IPv6: only the forms that need no expansion are told apart; "::" variants of zero fold
    bool zero = true;
    for (char c : s) if (c != ':' && c != '0') zero = false;
    return zero ? "::" : s;
  }
  int parts = 0, val = -1;
  bool ok = !s.empty();
  std::string out;
  for (size_t i = 0; i <= s.size() && ok; ++i) {
    if (i == s.size() || s[i] == '.') { if (val < 0 || val > 255) ok = false; else { out += (parts ? "." : "") + std::to_string(val); parts++; val = -1; } }
    else if (s[i] >= '0' && s[i] <= '9') val = (val < 0 ? 0 : val * 10) + (s[i] - '0');
    else ok = false;
  }
  return ok && parts == 4 ? out : "<nil>";
}
// HostPortUsage.Conflicts — hostportusage.go:76-87 (the pod being placed is never among the users here)
inline bool host_ports_conflict(const std::vector<HostPort>& wanted, const std::vector<HostPort>& used) {
  for (auto& w : wanted) for (auto& u : used) if (w.matches(u)) return true;
  return false;
}

struct Pod {
  std::string uid, name;
  Sym uid_s = kNoSym;                 // the UID as a symbol: what owner sets, caches and the queue's lastLen are keyed by
  Sym ns = kNoSym;
  std::vector<HostPort> host_ports;   // GetHostPorts — hostportusage.go:93-117 (hostIP "" reads 0.0.0.0)
  SymMap labels;
  long long creation = 0;
  Sym phase = W().Pending;
  Sym node_name = W().empty;
  ResourceList requests;
  SymMap node_selector;
  bool has_node_affinity = false;
  bool has_required = false;  // RequiredDuringSchedulingIgnoredDuringExecution != nil
  std::vector<NodeSelectorTerm> required_terms;
  std::vector<PreferredSchedulingTerm> preferred_terms;
  std::vector<Toleration> tolerations;
  std::vector<TopologySpreadConstraint> tscs;
  bool has_pod_affinity = false, has_pod_anti_affinity = false;
  std::vector<PodAffinityTerm> affinity_required, anti_required;
  std::vector<WeightedPodAffinityTerm> affinity_preferred, anti_preferred;
  bool owned_by_daemonset = false, owned_by_node = false;
  int input_index = 0;
  // volumeReqsByPod[pod.UID] (scheduler.go:138, :572): the alternatives VolumeTopology.GetRequirements derived from the pod's
  // volumes, one requirement set per valid combination of volume topologies; empty = no volume constraint
  std::vector<std::vector<NodeSelectorExpr>> volume_requirements;
  // scheduling.GetVolumes(pod) (volumeusage.go:83-114, scheduler.go:622-626): CSI driver -> the PVCs the pod mounts through it;
  // resolved from PVC / PV / StorageClass objects upstream of Solve()
  std::map<Sym, SymSet> volumes;
};

struct Offering {
  Requirements reqs;
  double price = 0;
  bool available = true;
  int reservation_capacity = 0;
  ResourceList capacity_override;            // Offering.CapacityOverride (types.go:484)
  bool has_overhead_override = false;        // Offering.OverheadOverride != nil
  ResourceList overhead_override;            // ... its Total()
  Sym capacity_type() const { return reqs.get(W().capacity_type).any(); }   // types.go:532
  Sym zone() const { return reqs.get(W().zone).any(); }                      // types.go:536
  Sym reservation_id() const { return reqs.get(W().reservation_id).any(); }  // types.go:540
};

struct AllocatableOfferings { ResourceList allocatable; std::vector<const Offering*> offerings; };

struct InstanceType {
  std::string name;
  Requirements reqs;
  std::vector<Offering> offerings;
  ResourceList capacity, overhead;
  std::vector<AllocatableOfferings> groups;  // AllocatableOfferingsList (types.go:325): base group first
  int catalog_index = 0;

  // computeAllocatable — types.go:271-294: lo.Assign replaces whole keys, then capacity - overhead, then the hugepage
  // reservation comes off memory (never below zero)
  ResourceList compute_allocatable(const ResourceList* cap_ov, const ResourceList* oh_ov) const {
    ResourceList cap = capacity, oh = overhead;
    if (cap_ov && !cap_ov->empty()) for (auto& kv : *cap_ov) cap[kv.first] = kv.second;
    if (oh_ov) for (auto& kv : *oh_ov) oh[kv.first] = kv.second;
    ResourceList a = res_subtract(cap, oh);
    for (auto& kv : cap) {
      if (str(kv.first).rfind("hugepages-", 0) == 0) {
        i128 cur = a.count(W().memory) ? a[W().memory] : 0;
        cur -= kv.second;
        if (cur < 0) cur = 0;
        a[W().memory] = cur;
      }
    }
    return a;
  }

  // precompute / groupOfferingsByOverride — types.go:202-269. Available offerings are grouped by their
  // (CapacityOverride, OverheadOverride) pair in first-seen order behind the base group; every group gets its own
  // allocatable. (Go keys the groups on the printed override; equal values printed differently would make two groups
  // with the same allocatable, which fits() — the only reader, nodeclaim.go:624-638 — cannot tell from one.)
  void precompute() {
    groups.clear();
    AllocatableOfferings base;
    base.allocatable = compute_allocatable(nullptr, nullptr);
    groups.push_back(base);
    std::vector<const Offering*> first;   // the offering whose overrides define group i (nullptr for the base)
    first.push_back(nullptr);
    for (auto& o : offerings) {
      if (!o.available) continue;
      if (o.capacity_override.empty() && !o.has_overhead_override) { groups[0].offerings.push_back(&o); continue; }
      size_t g = 1;
      for (; g < groups.size(); ++g)
        if (first[g]->capacity_override == o.capacity_override && first[g]->has_overhead_override == o.has_overhead_override &&
            first[g]->overhead_override == o.overhead_override) break;
      if (g == groups.size()) {
        AllocatableOfferings ng;
        ng.allocatable = compute_allocatable(&o.capacity_override, o.has_overhead_override ? &o.overhead_override : nullptr);
        groups.push_back(ng);
        first.push_back(&o);
      }
      groups[g].offerings.push_back(&o);
    }
  }
};

struct NodePool {
  std::string name;
  int weight = 0;
  std::vector<NodeSelectorExpr> requirements;
  SymMap labels;
  std::vector<Taint> taints;
  bool has_limits = false;
  ResourceList limits;
  bool is_static = false;
  Sym node_class_label_key = sym("karpenter.test.sh/testnodeclass"), node_class_name = sym("default");
  std::vector<int> instance_types;  // indices into the catalogue, in GetInstanceTypes order
};

// state.StateNode accessors the scheduler reads (existingnode.go:47-75, scheduler.go:792-858)
struct StateNode {
  std::string name;
  Sym name_s = kNoSym, hostname = kNoSym;
  SymMap labels;
  std::vector<Taint> taints;
  ResourceList available, capacity, daemonset_requests;
  std::vector<HostPort> host_ports;   // StateNode.HostPortUsage(): ports of the pods bound to the node (statenode.go:407,489)
  bool initialized = true, managed = true, has_node = true, marked_for_deletion = false;
  bool under_consolidate_after = false;  // disruption.IsUnderConsolidateAfter, evaluated upstream
  // StateNode.VolumeUsage() (statenode.go:411; volumeusage.go:178-189): the volumes of the pods bound to the node per CSI driver,
  // and the CSINode's per-driver attach limits
  std::map<Sym, SymSet> volumes;
  std::map<Sym, int> volume_limits;
};

// pods already bound in the cluster (topology.go:361-459 countDomains, :310-324 inverse anti-affinities)
struct ClusterPod {
  Pod pod;
};

struct Options {
  bool ignore_preferences = false;           // PreferencePolicyIgnore (scheduler.go:107)
  bool min_values_best_effort = false;       // MinValuesPolicyBestEffort (scheduler.go:117)
  bool reserved_capacity = false;            // FeatureGates.ReservedCapacity (nodeclaim.go:308)
  bool reserved_offering_strict = false;     // DisableReservedCapacityFallback (scheduler.go:103)
  bool enforce_consolidate_after = false;    // IsConsolidationSimulation (scheduler.go:123)
  long long max_steps = -1;                  // stand-in for the ctx deadline (scheduler.go:477): stop after N pops
  int truncate_instance_types = 0;           // > 0: Results.TruncateInstanceTypes(n) after Solve (scheduler.go:419-437)
  bool spot_to_spot_consolidation = false;   // FeatureGates.SpotToSpotConsolidation (consolidation.go:264), read by consolidation.hpp only
};

struct Problem {
  Options opts;
  std::vector<InstanceType> catalog;
  std::vector<NodePool> node_pools;
  std::vector<StateNode> state_nodes;
  std::vector<Pod> pods;
  std::vector<Pod> daemonset_pods;
  std::vector<Pod> cluster_pods;        // bound pods, for topology counting
  SymSet deleting_node_names;
  std::vector<std::pair<Sym, SymMap>> namespaces;   // the namespace lister: name, labels
};

// ------------------------------------------------------------------ JSON -> model
inline SymMap parse_strmap(const oj::Value& v) {
  SymMap m;
  for (auto& kv : v.members()) m.set(sym(kv.first), sym(kv.second.s()));
  return m;
}
inline ResourceList parse_resources(const oj::Value& v) {
  ResourceList r;
  for (auto& kv : v.members()) {
    if (kv.second.kind == oj::Value::Str) r[sym(kv.first)] = parse_quantity(kv.second.str);
    else if (kv.second.kind == oj::Value::Num && kv.second.is_int) r[sym(kv.first)] = (i128)kv.second.inum * 1000000000;
    else throw std::runtime_error("resource quantity must be a string or integer");
  }
  return r;
}
inline NodeSelectorExpr parse_expr(const oj::Value& v) {
  NodeSelectorExpr e;
  e.key = sym(v.at("key").s());
  e.op = parse_op(v.at("operator").s());
  for (auto& x : v.at("values").items()) e.values.push_back(sym(x.s()));
  if (v.has("minValues") && !v.at("minValues").is_null()) e.min_values = (int)v.at("minValues").i();
  return e;
}
inline std::vector<Taint> parse_taints(const oj::Value& v) {
  std::vector<Taint> out;
  for (auto& t : v.items()) out.push_back({sym(t.at("key").s()), sym(t.at("value").s()), sym(t.at("effect").s())});
  return out;
}
inline LabelSelector parse_selector(const oj::Value& v) {
  LabelSelector s;
  if (v.is_null()) return s;
  s.is_nil = false;
  s.match_labels = parse_strmap(v.at("matchLabels"));
  for (auto& e : v.at("matchExpressions").items()) {
    SelectorExpr x;
    x.key = sym(e.at("key").s());
    const std::string op = e.at("operator").s();
    x.op_text = sym(op);
    x.op = op == "In" ? SelOp::In : op == "NotIn" ? SelOp::NotIn : op == "Exists" ? SelOp::Exists : op == "DoesNotExist" ? SelOp::DoesNotExist : SelOp::Invalid;
    for (auto& val : e.at("values").items()) x.values.insert(sym(val.s()));
    s.match_expressions.push_back(x);
  }
  return s;
}
inline PodAffinityTerm parse_affinity_term(const oj::Value& v) {
  PodAffinityTerm t;
  t.selector = parse_selector(v.at("labelSelector"));
  t.topology_key = sym(v.at("topologyKey").s());
  for (auto& n : v.at("namespaces").items()) t.namespaces.push_back(sym(n.s()));
  if (v.has("namespaceSelector")) t.namespace_selector = parse_selector(v.at("namespaceSelector"));
  if (!t.namespace_selector.is_nil && !t.namespace_selector.valid()) throw std::runtime_error("parsing selector: invalid namespaceSelector");  // topology.go:545-547
  return t;
}
inline Requirements exprs_to_requirements(const std::vector<NodeSelectorExpr>& exprs) {  // requirements.go:49-65
  Requirements r;
  for (auto& e : exprs) r.add(Requirement::make(e.key, e.op, e.min_values, e.values));
  return r;
}
inline std::vector<HostPort> parse_host_ports(const oj::Value& v) {
  std::vector<HostPort> out;
  if (v.is_null()) return out;
  for (auto& e : v.items()) {
    HostPort h;
    h.port = (int)e.at("port").i(0);
    if (h.port == 0) continue;                                  // hostportusage.go:97
    std::string ip = e.at("ip").s("");
    h.ip = sym(canonical_ip(ip.empty() ? "0.0.0.0" : ip));      // hostportusage.go:103-106
    h.protocol = sym(e.at("protocol").s("TCP"));
    out.push_back(h);
  }
  return out;
}
inline Pod parse_pod(const oj::Value& v, int idx) {
  Pod p;
  p.host_ports = parse_host_ports(v.at("hostPorts"));
  p.input_index = idx;
  p.uid = v.at("uid").s();
  p.uid_s = sym(p.uid);
  p.ns = sym(v.at("namespace").s("default"));
  p.name = v.at("name").s(p.uid);
  p.labels = parse_strmap(v.at("labels"));
  p.creation = v.at("creationTimestamp").i(0);
  p.phase = sym(v.at("phase").s("Pending"));
  p.node_name = sym(v.at("nodeName").s(""));
  p.requests = parse_resources(v.at("requests"));
  p.node_selector = parse_strmap(v.at("nodeSelector"));
  p.owned_by_daemonset = v.at("ownedByDaemonSet").boolean_or(false);
  p.owned_by_node = v.at("ownedByNode").boolean_or(false);
  for (auto& alt : v.at("volumeRequirements").items()) {
    std::vector<NodeSelectorExpr> exprs;
    for (auto& e : alt.items()) exprs.push_back(parse_expr(e));
    p.volume_requirements.push_back(exprs);
  }
  for (auto& vv : v.at("volumes").items()) p.volumes[sym(vv.at("driver").s())].insert(sym(vv.at("pvc").s()));
  const oj::Value& na = v.at("nodeAffinity");
  if (!na.is_null()) {
    p.has_node_affinity = true;
    if (na.has("required") && !na.at("required").is_null()) {
      p.has_required = true;
      for (auto& term : na.at("required").items()) {
        NodeSelectorTerm t;
        for (auto& e : term.items()) t.push_back(parse_expr(e));
        p.required_terms.push_back(t);
      }
    }
    for (auto& pt : na.at("preferred").items()) {
      PreferredSchedulingTerm t;
      t.weight = (int)pt.at("weight").i();
      for (auto& e : pt.at("matchExpressions").items()) t.preference.push_back(parse_expr(e));
      p.preferred_terms.push_back(t);
    }
  }
  for (auto& t : v.at("tolerations").items())
    p.tolerations.push_back({sym(t.at("key").s()), sym(t.at("operator").s()), sym(t.at("value").s()), sym(t.at("effect").s())});
  for (auto& c : v.at("topologySpreadConstraints").items()) {
    TopologySpreadConstraint t;
    t.max_skew = (int)c.at("maxSkew").i(1);
    t.topology_key = sym(c.at("topologyKey").s());
    t.when_unsatisfiable = sym(c.at("whenUnsatisfiable").s("DoNotSchedule"));
    t.selector = parse_selector(c.at("labelSelector"));
    if (c.has("minDomains") && !c.at("minDomains").is_null()) t.min_domains = (int)c.at("minDomains").i();
    if (c.has("nodeTaintsPolicy") && !c.at("nodeTaintsPolicy").is_null()) t.node_taints_policy = sym(c.at("nodeTaintsPolicy").s());
    if (c.has("nodeAffinityPolicy") && !c.at("nodeAffinityPolicy").is_null()) t.node_affinity_policy = sym(c.at("nodeAffinityPolicy").s());
    for (auto& k : c.at("matchLabelKeys").items()) t.match_label_keys.push_back(sym(k.s()));
    p.tscs.push_back(t);
  }
  const oj::Value& pa = v.at("podAffinity");
  if (!pa.is_null()) {
    p.has_pod_affinity = true;
    for (auto& t : pa.at("required").items()) p.affinity_required.push_back(parse_affinity_term(t));
    for (auto& t : pa.at("preferred").items()) p.affinity_preferred.push_back({(int)t.at("weight").i(), parse_affinity_term(t.at("term"))});
  }
  const oj::Value& paa = v.at("podAntiAffinity");
  if (!paa.is_null()) {
    p.has_pod_anti_affinity = true;
    for (auto& t : paa.at("required").items()) p.anti_required.push_back(parse_affinity_term(t));
    for (auto& t : paa.at("preferred").items()) p.anti_preferred.push_back({(int)t.at("weight").i(), parse_affinity_term(t.at("term"))});
  }
  return p;
}

// Deterministic expansion of {"count": n, "uidSeed": s, "template": pod} groups: pod i gets a 128-bit hex uid from
// splitmix64(seed, i). Both the oracle and the product host expand groups with this same public recipe so that
// million-pod problems do not need million-entry JSON files.
inline uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
inline std::string group_uid(uint64_t seed, uint64_t i) {
  uint64_t st = seed * 0x9E3779B97F4A7C15ULL + i * 0xD1B54A32D192ED03ULL + 0x2545F4914F6CDD1DULL;
  uint64_t a = splitmix64(st), b = splitmix64(st);
  char buf[40];
  snprintf(buf, sizeof buf, "%08x-%04x-%04x-%04x-%012llx", (unsigned)(a >> 32), (unsigned)((a >> 16) & 0xffff),
           (unsigned)(a & 0xffff), (unsigned)(b >> 48), (unsigned long long)(b & 0xffffffffffffULL));
  return buf;
}

inline Problem parse_problem(const oj::Value& root) {
  Problem pr;
  auto& reg = labels_registry();
  reg = Labels();
  for (auto& k : root.at("wellKnownLabels").items()) reg.well_known.insert(sym(k.s()));
  const oj::Value& o = root.at("options");
  pr.opts.ignore_preferences = o.at("preferencePolicy").s("Respect") == "Ignore";
  pr.opts.min_values_best_effort = o.at("minValuesPolicy").s("Strict") == "BestEffort";
  pr.opts.reserved_capacity = o.at("reservedCapacity").boolean_or(false);
  pr.opts.reserved_offering_strict = o.at("reservedOfferingMode").s("Fallback") == "Strict";
  pr.opts.enforce_consolidate_after = o.at("consolidationSimulation").boolean_or(false);
  pr.opts.max_steps = o.at("maxSteps").i(-1);
  pr.opts.truncate_instance_types = (int)o.at("truncateInstanceTypes").i(0);
  pr.opts.spot_to_spot_consolidation = o.at("spotToSpotConsolidation").boolean_or(false);

  std::map<std::string, int> it_index;
  int ci = 0;
  for (auto& v : root.at("instanceTypes").items()) {
    InstanceType it;
    it.catalog_index = ci++;
    it.name = v.at("name").s();
    std::vector<NodeSelectorExpr> ex;
    for (auto& e : v.at("requirements").items()) ex.push_back(parse_expr(e));
    it.reqs = exprs_to_requirements(ex);
    it.capacity = parse_resources(v.at("capacity"));
    it.overhead = parse_resources(v.at("overhead"));
    for (auto& of : v.at("offerings").items()) {
      Offering off;
      std::vector<NodeSelectorExpr> oe;
      for (auto& e : of.at("requirements").items()) oe.push_back(parse_expr(e));
      off.reqs = exprs_to_requirements(oe);
      off.price = of.at("price").d();
      off.available = of.at("available").boolean_or(true);
      off.reservation_capacity = (int)of.at("reservationCapacity").i(0);
      if (of.has("capacityOverride") && !of.at("capacityOverride").is_null()) off.capacity_override = parse_resources(of.at("capacityOverride"));
      if (of.has("overheadOverride") && !of.at("overheadOverride").is_null()) { off.has_overhead_override = true; off.overhead_override = parse_resources(of.at("overheadOverride")); }
      it.offerings.push_back(off);
    }
    it_index[it.name] = (int)pr.catalog.size();
    pr.catalog.push_back(std::move(it));
  }
  for (auto& it : pr.catalog) it.precompute();

  for (auto& v : root.at("nodePools").items()) {
    NodePool np;
    np.name = v.at("name").s();
    np.weight = (int)v.at("weight").i(0);
    for (auto& e : v.at("requirements").items()) np.requirements.push_back(parse_expr(e));
    np.labels = parse_strmap(v.at("labels"));
    np.taints = parse_taints(v.at("taints"));
    if (v.has("limits") && !v.at("limits").is_null()) { np.has_limits = true; np.limits = parse_resources(v.at("limits")); }
    np.is_static = v.at("static").boolean_or(false);
    if (v.has("nodeClassLabelKey")) np.node_class_label_key = sym(v.at("nodeClassLabelKey").s());
    if (v.has("nodeClassName")) np.node_class_name = sym(v.at("nodeClassName").s());
    if (v.has("instanceTypes") && !v.at("instanceTypes").is_null()) {
      for (auto& n : v.at("instanceTypes").items()) {
        auto f = it_index.find(n.s());
        if (f == it_index.end()) throw std::runtime_error("unknown instance type " + n.s());
        np.instance_types.push_back(f->second);
      }
    } else {
      for (int i = 0; i < (int)pr.catalog.size(); ++i) np.instance_types.push_back(i);
    }
    pr.node_pools.push_back(np);
  }
  for (auto& v : root.at("stateNodes").items()) {
    StateNode n;
    n.name = v.at("name").s();
    n.name_s = sym(n.name);
    n.labels = parse_strmap(v.at("labels"));
    n.hostname = v.has("hostname") ? sym(v.at("hostname").s()) : (n.labels.count(W().hostname) ? n.labels.find(W().hostname)->second : n.name_s);
    n.taints = parse_taints(v.at("taints"));
    n.available = parse_resources(v.at("available"));
    n.capacity = parse_resources(v.at("capacity"));
    n.daemonset_requests = parse_resources(v.at("daemonSetRequests"));
    n.host_ports = parse_host_ports(v.at("hostPorts"));
    n.initialized = v.at("initialized").boolean_or(true);
    n.managed = v.at("managed").boolean_or(true);
    n.has_node = v.at("hasNode").boolean_or(true);
    n.marked_for_deletion = v.at("markedForDeletion").boolean_or(false);
    n.under_consolidate_after = v.at("underConsolidateAfter").boolean_or(false);
    for (auto& vv : v.at("volumeUsage").at("volumes").items()) n.volumes[sym(vv.at("driver").s())].insert(sym(vv.at("pvc").s()));
    for (auto& kv : v.at("volumeUsage").at("limits").members()) n.volume_limits[sym(kv.first)] = (int)kv.second.i();
    pr.state_nodes.push_back(n);
  }
  for (auto& n : root.at("deletingNodeNames").items()) pr.deleting_node_names.insert(sym(n.s()));
  int idx = 0;
  for (auto& v : root.at("pods").items()) pr.pods.push_back(parse_pod(v, idx++));
  for (auto& g : root.at("podGroups").items()) {
    Pod tmpl = parse_pod(g.at("template"), 0);
    uint64_t seed = (uint64_t)g.at("uidSeed").i(0);
    long long n = g.at("count").i(0);
    for (long long i = 0; i < n; ++i) {
      Pod p = tmpl;
      p.uid = group_uid(seed, (uint64_t)i);
      p.uid_s = sym(p.uid);
      p.name = p.uid;
      p.input_index = idx++;
      pr.pods.push_back(std::move(p));
    }
  }
  int di = 0;
  for (auto& v : root.at("daemonSetPods").items()) pr.daemonset_pods.push_back(parse_pod(v, di++));
  int cpi = 0;
  for (auto& v : root.at("clusterPods").items()) pr.cluster_pods.push_back(parse_pod(v, cpi++));
  for (auto& v : root.at("namespaces").items()) pr.namespaces.push_back({sym(v.at("name").s()), parse_strmap(v.at("labels"))});
  return pr;
}

}  // namespace oracle
