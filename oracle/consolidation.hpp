// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// Restatement of the consolidation DECISION that wraps a simulation (round 4): what the reference does with the Results of
// SimulateScheduling. Until now the oracle restated Solve() only and the verdicts of the product's sweeps (ksched_sweep, C++)
// were checked against the product's own Python (karpenter_amd/disruption.py) — two builder-written twins. This file is the
// independent third: it works on the oracle's own objects (real requirement sets, real instance-type lists), shares no line
// with either, and follows
//   pkg/controllers/disruption/helpers.go:53-155            SimulateScheduling (the part after Solve: truncation, uninitialized nodes)
//   pkg/controllers/provisioning/scheduling/scheduler.go:388-392, :419-437   AllNonPendingPodsScheduled, TruncateInstanceTypes
//   pkg/controllers/disruption/consolidation.go:159-256     computeConsolidation
//   pkg/controllers/disruption/consolidation.go:261-343     computeSpotToSpotConsolidation
//   pkg/controllers/provisioning/scheduling/nodeclaim.go:411-420   RemoveInstanceTypeOptionsByPriceAndMinValues
//   pkg/cloudprovider/types.go:298-305, :357-365, :547-598  OfferingPrice, InstanceTypes.Compatible, Offerings.*
//   pkg/controllers/disruption/types.go:113-127, :161-211   resolveNodePrice, NewCandidate (instance type / capacity type of a node)
//   pkg/controllers/disruption/multinodeconsolidation.go:147-158, :209-246   the replace check of the binary search, filterOutSameInstanceType
//   pkg/utils/pod/scheduling.go:102-108                      IsProvisionable
#pragma once
#include <cmath>

#include "scheduler.hpp"

namespace oracle {

enum class Decision { NoOp = 0, Delete = 1, Replace = 2 };

struct Command {   // disruption.Command, the fields the sweep's callers read
  Decision decision = Decision::NoOp;
  std::vector<const InstanceType*> replacement;   // Replacements[0].InstanceTypeOptions, in the order the reference leaves them
  Requirements replacement_reqs;                   // Replacements[0].Requirements
  std::string reason;                              // the Unconsolidatable event text ("" when none is published)
  bool pinned_to_spot = false;                     // consolidation.go:238-243 added capacity-type In [spot] to the replacement
};

// Candidate — disruption/types.go:76-92, built by NewCandidate :161-211 from the state node
struct Candidate {
  const StateNode* node = nullptr;
  const InstanceType* instance_type = nullptr;   // instanceTypeMap[node.Labels()[LabelInstanceTypeStable]]; nil when unknown
  Sym capacity_type = kNoSym;                    // node.Labels()[CapacityTypeLabelKey]
  double price = 0;                              // resolveNodePrice
};

// InstanceType.OfferingPrice — types.go:298-305 (every offering, available or not; first match)
inline bool offering_price(const InstanceType& it, Sym zone, Sym capacity_type, double& price) {
  for (auto& o : it.offerings)
    if (o.zone() == zone && o.capacity_type() == capacity_type) { price = o.price; return true; }
  return false;
}
// resolveNodePrice — disruption/types.go:113-127
inline double resolve_node_price(const StateNode& node, const InstanceType* it) {
  if (!it) return 0;
  auto* z = node.labels.find(W().zone);
  auto* c = node.labels.find(W().capacity_type);
  double price;
  if (!offering_price(*it, z ? z->second : W().empty, c ? c->second : W().empty, price)) return 0;
  if (std::isnan(price)) return 0;
  return price;
}
inline Candidate make_candidate(const Problem& pr, const StateNode& node) {
  Candidate c;
  c.node = &node;
  auto* itl = node.labels.find(W().instance_type);
  if (itl) for (auto& it : pr.catalog) if (it.name == str(itl->second)) { c.instance_type = &it; break; }
  auto* ctl = node.labels.find(W().capacity_type);
  c.capacity_type = ctl ? ctl->second : W().empty;
  c.price = resolve_node_price(node, c.instance_type);
  return c;
}

// pod.IsProvisionable — pkg/utils/pod/scheduling.go:102-108. FailedToSchedule / IsPreempting are conditions of the pods
// GetPendingPods already selected upstream (helpers.go:72); what is left to tell apart here is bound vs not, and ownership.
inline bool is_provisionable(const Pod& p) { return p.node_name == W().empty && !p.owned_by_daemonset && !p.owned_by_node; }

// Results.TruncateInstanceTypes — scheduler.go:419-437 ; InstanceTypes.Truncate — types.go:437-449
inline void truncate_instance_types(const Problem& pr, Results& res, int max_items) {
  std::vector<NodeClaim*> valid;
  for (auto* nc : res.new_node_claims) {
    order_by_price(nc->its, nc->reqs);
    std::vector<const InstanceType*> cut(nc->its.begin(), nc->its.begin() + std::min<size_t>(nc->its.size(), (size_t)max_items));
    bool ok = true;
    if (nc->reqs.has_min_values() && !pr.opts.min_values_best_effort) satisfies_min_values(cut, nc->reqs, nullptr, &ok);
    if (!ok) { for (auto* p : nc->pods) res.pod_errors[p->uid] = {ERR_MIN_VALUES, 128}; continue; }
    nc->its = cut;
    valid.push_back(nc);
  }
  res.new_node_claims = valid;
}

// helpers.go:133-153: a pod the simulation put on a node that is not initialized yet makes the decision unsafe — unless the
// pod comes from a node that is already deleting
inline void mark_uninitialized_nodes(const Problem& pr, Results& res) {
  for (auto* en : res.existing_nodes) {
    if (en->node->initialized) continue;
    for (auto* p : en->pods)
      if (!(p->node_name != W().empty && pr.deleting_node_names.count(p->node_name))) res.pod_errors[p->uid] = {100, 0};   // UninitializedNodeError
  }
}
// Results.AllNonPendingPodsScheduled — scheduler.go:388-392
inline bool all_non_pending_pods_scheduled(const std::vector<Pod>& pods, const Results& res) {
  if (res.pod_errors.empty()) return true;
  for (auto& p : pods) if (res.pod_errors.count(p.uid) && !is_provisionable(p)) return false;
  return true;
}

// NodeClaim.RemoveInstanceTypeOptionsByPriceAndMinValues — nodeclaim.go:411-420 ; false = the minValues error
inline bool remove_by_price_and_min_values(std::vector<const InstanceType*>& its, const Requirements& reqs, double max_price) {
  std::vector<const InstanceType*> kept;
  for (auto* it : its) if (worst_launch_price(*it, reqs) < max_price) kept.push_back(it);   // Offerings.Available().WorstLaunchPrice
  its = kept;
  bool ok;
  satisfies_min_values(its, reqs, nullptr, &ok);
  return ok;
}

// computeSpotToSpotConsolidation — consolidation.go:261-343
inline Command spot_to_spot(const Problem& pr, const std::vector<Candidate>& candidates, NodeClaim& claim, double candidate_price) {
  Command cmd;
  if (!pr.opts.spot_to_spot_consolidation) {
    if (candidates.size() == 1) cmd.reason = "SpotToSpotConsolidation is disabled, can't replace a spot node with a spot node";
    return cmd;
  }
  claim.reqs.add(Requirement::make(W().capacity_type, Op::In, W().spot));
  cmd.pinned_to_spot = true;
  // InstanceTypes.Compatible — types.go:357-365
  std::vector<const InstanceType*> compatible;
  for (auto* it : claim.its) {
    bool has = false;
    for (auto& o : it->offerings) if (o.available && offering_compatible(claim.reqs, o)) { has = true; break; }
    if (has) compatible.push_back(it);
  }
  claim.its = compatible;
  if (!remove_by_price_and_min_values(claim.its, claim.reqs, candidate_price)) {
    if (candidates.size() == 1) cmd.reason = "Filtering by price: minValues requirement is not met";
    return cmd;
  }
  if (claim.its.empty()) {
    if (candidates.size() == 1) cmd.reason = "Can't replace with a cheaper node";
    return cmd;
  }
  const size_t kMin = 15;   // MinInstanceTypesForSpotToSpotConsolidation, consolidation.go:46
  if (candidates.size() == 1) {
    if (claim.its.size() < kMin) {
      cmd.reason = "SpotToSpotConsolidation requires " + std::to_string(kMin) + " cheaper instance type options than the current candidate to consolidate, got " + std::to_string(claim.its.size());
      return cmd;
    }
    size_t keep = kMin;
    if (claim.reqs.has_min_values()) {
      bool ok;
      int need = satisfies_min_values(claim.its, claim.reqs, nullptr, &ok);
      keep = std::max(kMin, (size_t)need);
    }
    if (claim.its.size() > keep) claim.its.resize(keep);
  }
  cmd.decision = Decision::Replace;
  cmd.replacement = claim.its;
  cmd.replacement_reqs = claim.reqs;
  return cmd;
}

// computeConsolidation — consolidation.go:159-256, from the finished simulation (`res` after truncate_instance_types and
// mark_uninitialized_nodes). Narrows the one NodeClaim of `res` in place, as the reference does.
inline Command compute_consolidation(const Problem& pr, const std::vector<Candidate>& candidates, const std::vector<Pod>& pods, Results& res) {
  Command cmd;
  if (!all_non_pending_pods_scheduled(pods, res)) {
    if (candidates.size() == 1) cmd.reason = "not all pods would schedule";
    return cmd;
  }
  if (res.new_node_claims.empty()) { cmd.decision = Decision::Delete; return cmd; }
  if (res.new_node_claims.size() != 1) {
    if (candidates.size() == 1) cmd.reason = "Can't remove without creating " + std::to_string(res.new_node_claims.size()) + " candidates";
    return cmd;
  }
  double candidate_price = 0;   // sumCandidatePrices — balanced.go:185-187
  for (auto& c : candidates) candidate_price += c.price;
  bool all_existing_are_spot = true;
  for (auto& c : candidates) if (c.capacity_type != W().spot) all_existing_are_spot = false;
  NodeClaim& claim = *res.new_node_claims[0];
  order_by_price(claim.its, claim.reqs);
  if (all_existing_are_spot && claim.reqs.get(W().capacity_type).has(W().spot)) return spot_to_spot(pr, candidates, claim, candidate_price);
  if (!remove_by_price_and_min_values(claim.its, claim.reqs, candidate_price)) {
    if (candidates.size() == 1) cmd.reason = "Filtering by price: minValues requirement is not met";
    return cmd;
  }
  if (claim.its.empty()) {
    if (candidates.size() == 1) cmd.reason = "Can't replace with a cheaper node";
    return cmd;
  }
  Requirement ct = claim.reqs.get(W().capacity_type);
  if (ct.has(W().spot) && ct.has(W().on_demand)) { claim.reqs.add(Requirement::make(W().capacity_type, Op::In, W().spot)); cmd.pinned_to_spot = true; }
  cmd.decision = Decision::Replace;
  cmd.replacement = claim.its;
  cmd.replacement_reqs = claim.reqs;
  return cmd;
}

// filterOutSameInstanceType — multinodeconsolidation.go:209-246 ; false = the error of the minValues re-check
inline bool filter_out_same_instance_type(Command& cmd, const std::vector<Candidate>& consolidate) {
  std::set<std::string> existing;
  std::map<std::string, double> prices;   // a missing entry reads 0, like the Go map
  for (auto& c : consolidate) {
    if (!c.instance_type) continue;   // (the reference would dereference nil; candidates without a known type are filtered upstream, types.go:179-186)
    existing.insert(c.instance_type->name);
    Requirements node_reqs = label_requirements(c.node->labels);
    const Offering* cheapest = nullptr;   // Offerings.Compatible(...).Cheapest(): lo.MinBy keeps the first minimum
    for (auto& o : c.instance_type->offerings)
      if (node_reqs.compatible(o.reqs, true) && (!cheapest || o.price < cheapest->price)) cheapest = &o;
    if (!cheapest) continue;
    auto f = prices.find(c.instance_type->name);
    double existing_price = f == prices.end() ? DBL_MAX : f->second;
    if (cheapest->price < existing_price) prices[c.instance_type->name] = cheapest->price;
  }
  double max_price = DBL_MAX;
  for (auto* it : cmd.replacement)
    if (existing.count(it->name)) {
      auto f = prices.find(it->name);
      double p = f == prices.end() ? 0.0 : f->second;
      if (p < max_price) max_price = p;
    }
  return remove_by_price_and_min_values(cmd.replacement, cmd.replacement_reqs, max_price);
}

// One step of firstNConsolidationOption's search (multinodeconsolidation.go:140-158): the command of a candidate prefix and
// whether the search would accept it (before the evaluator, which scores float costs outside this path).
inline Command multi_node_step(const Problem& pr, const std::vector<Candidate>& candidates, const std::vector<Pod>& pods, Results& res) {
  Command cmd = compute_consolidation(pr, candidates, pods, res);
  if (cmd.decision == Decision::Replace) {
    const bool ok = filter_out_same_instance_type(cmd, candidates);
    if (!ok || cmd.replacement.empty()) {
      cmd.decision = Decision::NoOp;
      cmd.reason = ok ? "every replacement option is one of the types being removed, or more expensive" : "minValues requirement is not met after the same-type filter";
      cmd.replacement.clear();
    }
  }
  return cmd;
}

}  // namespace oracle
