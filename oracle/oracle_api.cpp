// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// C entry points for the CPU restatement of the reference Solve() path. Imported only by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg. PARITY STATUS: the algebra is pinned against the
// reference's own truth tables (tests/golden/, transcribed from pkg/scheduling/*_test.go); the tie order of Go's
// sort.Slice (pdqsort.hpp) and Toleration.ToleratesTaint are restated third-party code and are UNPINNED here (no Go
// toolchain in this image); topology tie-breaks are canonicalised because the reference itself leaves them to Go
// map iteration order.
#include <chrono>
#include <thread>
#include <cstring>

#include "consolidation.hpp"

using namespace oracle;

static std::string i128_to_string(i128 v) {
  if (v == 0) return "0";
  bool neg = v < 0;
  unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
  std::string s;
  while (u) { s += (char)('0' + (int)(u % 10)); u /= 10; }
  if (neg) s += '-';
  std::reverse(s.begin(), s.end());
  return s;
}

static oj::Value req_to_json(const Requirement& r) {
  oj::Value o = oj::Value::object();
  o.set("key", oj::Value::string(str(r.key)));
  o.set("complement", oj::Value::boolean(r.complement));
  oj::Value vals = oj::Value::array();
  for (auto& v : r.values.strings()) vals.push(oj::Value::string(v));   // in string order
  o.set("values", vals);
  o.set("gte", r.gte ? oj::Value::integer(*r.gte) : oj::Value());
  o.set("lte", r.lte ? oj::Value::integer(*r.lte) : oj::Value());
  o.set("minValues", r.min_values ? oj::Value::integer(*r.min_values) : oj::Value());
  o.set("operator", oj::Value::string(op_name(r.op())));
  return o;
}
static Requirement req_from_json(const oj::Value& v) {
  std::vector<std::string> vals;
  for (auto& x : v.at("values").items()) vals.push_back(x.s());
  std::optional<int> mv;
  if (v.has("minValues") && !v.at("minValues").is_null()) mv = (int)v.at("minValues").i();
  return Requirement::make_s(v.at("key").s(), parse_op(v.at("operator").s()), mv, vals);
}
static Requirements reqs_from_json(const oj::Value& v) {
  Requirements r;
  for (auto& x : v.items()) r.add(req_from_json(x));
  return r;
}
static oj::Value reqs_to_json(const Requirements& r) {
  oj::Value a = oj::Value::array();
  std::vector<const Requirement*> by_key;
  for (auto& q : r.m) by_key.push_back(&q);
  std::sort(by_key.begin(), by_key.end(), [](const Requirement* x, const Requirement* y) { return str(x->key) < str(y->key); });
  for (auto* q : by_key) a.push(req_to_json(*q));
  return a;
}
static oj::Value res_to_json(const ResourceList& r) {
  oj::Value o = oj::Value::object();
  std::vector<const ResEntry*> by_name;
  for (auto& kv : r) by_name.push_back(&kv);
  std::sort(by_name.begin(), by_name.end(), [](const ResEntry* x, const ResEntry* y) { return str(x->first) < str(y->first); });
  for (auto* kv : by_name) o.set(str(kv->first), oj::Value::string(i128_to_string(kv->second)));  // nano-units, decimal string
  return o;
}
static char* dup_out(const oj::Value& v) {
  std::string s;
  oj::write(v, s);
  char* out = (char*)malloc(s.size() + 1);
  memcpy(out, s.c_str(), s.size() + 1);
  return out;
}
static char* err_out(const std::string& m) {
  oj::Value o = oj::Value::object();
  o.set("error", oj::Value::string(m));
  return dup_out(o);
}

extern "C" {

void oracle_free(char* p) { free(p); }

// parse + write of one document: lets a test hold this parser against an independent one (Python's json) — the product's host
// library carries a copy of the same parser, so a parse bug would otherwise be common to checker and product
char* oracle_json_roundtrip(const char* doc) {
  try {
    return dup_out(oj::Parser(doc).parse());
  } catch (const std::exception& e) {
    return err_out(e.what());
  }
}

// Full Solve(): problem JSON -> results JSON.
struct ProbeVerdict { bool want = false, multi_node = false; const std::vector<size_t>* candidates = nullptr; };   // candidates: state-node positions, in the caller's order
static oj::Value solve_doc(const Problem& pr, const std::vector<Pod>* probe_pods = nullptr, const std::vector<char>* removed = nullptr, const ProbeVerdict* verdict = nullptr);
char* oracle_solve_json(const char* problem_json) {
  try {
    oj::Value root = oj::Parser(problem_json).parse();
    Problem pr = parse_problem(root);
    return dup_out(solve_doc(pr));
  } catch (const std::exception& e) {
    return err_out(e.what());
  }
}

// SimulateScheduling (disruption/helpers.go:53-155) for many candidate sets of ONE cluster: {"problem": the cluster as a problem
// document (every node a state node, no pods), "probes": [{"removeNodes": [names], "pods": [pod documents]}], "threads": n} ->
// {"results": [one Results document per probe]}. Every probe is a fresh Scheduler over the cluster without its candidates —
// what the reference does per simulation; only the parse of the cluster document is shared. The simulations are independent,
// so `threads` of them run at a time (the reference's own candidate fan-out is parallelizeUntil, scheduler.go:939-961).
// "verdicts": true adds to every Results document "verdict": the command computeConsolidation derives from that simulation
// (consolidation.hpp; candidates = removeNodes in the order given) — with "multiNode": true the command of one step of
// firstNConsolidationOption's search, i.e. a replacement also has to survive filterOutSameInstanceType.
char* oracle_sweep_json(const char* doc_json) {
  try {
    oj::Value root = oj::Parser(doc_json).parse();
    const Problem pr = parse_problem(root.at("problem"));
    std::map<std::string, size_t> by_name;
    for (size_t i = 0; i < pr.state_nodes.size(); ++i) by_name[pr.state_nodes[i].name] = i;
    const auto& probes = root.at("probes").items();
    const size_t n = probes.size();
    std::vector<std::vector<Pod>> pods(n);
    std::vector<std::vector<char>> removed(n);
    std::vector<std::vector<size_t>> cand(n);
    const bool want_verdicts = root.at("verdicts").boolean_or(false), multi_node = root.at("multiNode").boolean_or(false);
    for (size_t i = 0; i < n; ++i) {
      removed[i].assign(pr.state_nodes.size(), 0);
      for (auto& nn : probes[i].at("removeNodes").items()) {
        auto f = by_name.find(nn.s());
        if (f == by_name.end()) throw std::runtime_error("probe removes an unknown node " + nn.s());
        if (!removed[i][f->second]) cand[i].push_back(f->second);
        removed[i][f->second] = 1;
      }
      int idx = 0;
      for (auto& v : probes[i].at("pods").items()) pods[i].push_back(parse_pod(v, idx++));
    }
    std::vector<oj::Value> docs(n);
    std::vector<std::string> errors(n);
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(n, (size_t)root.at("threads").i(1)));
    auto work = [&](size_t t) {
      for (size_t i = t; i < n; i += n_threads) {
        ProbeVerdict pv;
        pv.want = want_verdicts; pv.multi_node = multi_node; pv.candidates = &cand[i];
        try { docs[i] = solve_doc(pr, &pods[i], &removed[i], &pv); } catch (const std::exception& e) { errors[i] = e.what(); }
      }
    };
    if (n_threads == 1) work(0);
    else { std::vector<std::thread> pool; for (size_t t = 0; t < n_threads; ++t) pool.emplace_back(work, t); for (auto& th : pool) th.join(); }
    oj::Value results = oj::Value::array();
    for (size_t i = 0; i < n; ++i) { if (!errors[i].empty()) throw std::runtime_error(errors[i]); results.push(docs[i]); }
    oj::Value out = oj::Value::object();
    out.set("results", results);
    return dup_out(out);
  } catch (const std::exception& e) {
    return err_out(e.what());
  }
}

static oj::Value solve_doc(const Problem& pr, const std::vector<Pod>* probe_pods, const std::vector<char>* removed, const ProbeVerdict* verdict) {
  {
    Scheduler s;
    auto t0 = std::chrono::steady_clock::now();
    // ORACLE_THREADS / ORACLE_PAR_MIN: candidate fan-out of the in-flight scan (parallelizeUntil, scheduler.go:939-961) for the
    // offline pins and the N-thread CPU baseline; the Results and counters are those of the sequential scan
    if (const char* t = getenv("ORACLE_THREADS")) s.threads = std::max(1, atoi(t));
    if (const char* t = getenv("ORACLE_PROGRESS")) s.progress_every = atoll(t);
    if (const char* t = getenv("ORACLE_PAR_MIN")) s.parallel_min = (size_t)std::max(1, atoi(t));
    s.init(pr, probe_pods, removed);
    auto t1 = std::chrono::steady_clock::now();
    Results res = s.solve();
    auto t2 = std::chrono::steady_clock::now();
    if (getenv("ORACLE_TIMING")) fprintf(stderr, "oracle timing: init %.1f s, solve %.1f s (sort.Slice %.1f s, in-flight scan %.1f s), %d thread(s)\n", std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), s.t_sort, s.t_scan, s.threads);

    oj::Value out = oj::Value::object();
    oj::Value claims = oj::Value::array();
    double total_cost = 0;
    // Results.TruncateInstanceTypes — scheduler.go:419-437 ; InstanceTypes.Truncate — types.go:437-449
    if (pr.opts.truncate_instance_types > 0) truncate_instance_types(pr, res, pr.opts.truncate_instance_types);
    for (auto* nc : res.new_node_claims) {
      oj::Value c = oj::Value::object();
      c.set("nodePool", oj::Value::string(nc->tmpl->nodepool_name));
      c.set("hostname", oj::Value::string(str(nc->hostname)));
      oj::Value pods = oj::Value::array();
      for (auto* p : nc->pods) pods.push(oj::Value::string(p->uid));
      c.set("pods", pods);
      oj::Value its = oj::Value::array();
      double cheapest = DBL_MAX;
      for (auto* it : nc->its) {
        its.push(oj::Value::string(it->name));
        double p = min_compatible_price(*it, nc->reqs);
        if (p < cheapest) cheapest = p;
      }
      c.set("instanceTypes", its);
      c.set("requirements", reqs_to_json(nc->reqs));
      c.set("requests", res_to_json(nc->requests));
      oj::Value ann = oj::Value::object();
      for (auto& kv : nc->annotations) ann.set(kv.first, oj::Value::string(kv.second));
      c.set("annotations", ann);
      oj::Value ro = oj::Value::array();
      for (auto* o : nc->reserved_offerings) ro.push(oj::Value::string(str(o->reservation_id())));
      c.set("reservedOfferings", ro);
      c.set("cheapestPrice", oj::Value::number(cheapest == DBL_MAX ? -1.0 : cheapest));
      if (cheapest != DBL_MAX) total_cost += cheapest;
      claims.push(c);
    }
    out.set("newNodeClaims", claims);
    oj::Value ens = oj::Value::array();
    for (auto* en : res.existing_nodes) {
      oj::Value e = oj::Value::object();
      e.set("name", oj::Value::string(en->node->name));
      oj::Value pods = oj::Value::array();
      for (auto* p : en->pods) pods.push(oj::Value::string(p->uid));
      e.set("pods", pods);
      e.set("initialized", oj::Value::boolean(en->node->initialized));
      ens.push(e);
    }
    out.set("existingNodes", ens);
    oj::Value errs = oj::Value::object();
    for (auto& kv : res.pod_errors) {
      oj::Value e = oj::Value::object();
      e.set("code", oj::Value::integer(kv.second.first));
      e.set("diag", oj::Value::integer(kv.second.second));
      errs.set(kv.first, e);
    }
    out.set("podErrors", errs);
    out.set("timedOut", oj::Value::boolean(res.timed_out));
    out.set("packingCost", oj::Value::number(total_cost));
    oj::Value c = oj::Value::object();
    c.set("binEvaluations", oj::Value::integer(s.ctr.bin_evaluations));
    c.set("instanceTypeEvaluations", oj::Value::integer(s.ctr.it_evaluations));
    c.set("sorts", oj::Value::integer(s.ctr.sorts));
    c.set("pops", oj::Value::integer(s.ctr.pops));
    c.set("relaxations", oj::Value::integer(s.ctr.relaxations));
    c.set("pods", oj::Value::integer((long long)s.pods.size()));
    c.set("initSeconds", oj::Value::number(std::chrono::duration<double>(t1 - t0).count()));
    c.set("solveSeconds", oj::Value::number(std::chrono::duration<double>(t2 - t1).count()));
    out.set("counters", c);
    if (verdict && verdict->want) {
      // the decision on top of this simulation (consolidation.hpp); computed last: it narrows the NodeClaim it judges
      mark_uninitialized_nodes(pr, res);
      std::vector<Candidate> cands;
      for (size_t e : *verdict->candidates) cands.push_back(make_candidate(pr, pr.state_nodes[e]));
      Command cmd = verdict->multi_node && cands.size() > 1 ? multi_node_step(pr, cands, s.pods, res) : compute_consolidation(pr, cands, s.pods, res);
      oj::Value v = oj::Value::object();
      v.set("decision", oj::Value::string(cmd.decision == Decision::Delete ? "delete" : cmd.decision == Decision::Replace ? "replace" : "no-op"));
      oj::Value names = oj::Value::array();
      for (auto* it : cmd.replacement) names.push(oj::Value::string(it->name));
      v.set("replacement", cmd.decision == Decision::Replace ? names : oj::Value());
      Requirement ct = cmd.replacement_reqs.get(W().capacity_type);
      v.set("replacementCapacityTypes", cmd.decision == Decision::Replace && ct.op() == Op::In ? [&] { oj::Value a = oj::Value::array(); for (auto& x : ct.values.strings()) a.push(oj::Value::string(x)); return a; }() : oj::Value());
      v.set("pinnedToSpot", oj::Value::boolean(cmd.pinned_to_spot));
      v.set("reason", oj::Value::string(cmd.reason));
      v.set("allNonPendingPodsScheduled", oj::Value::boolean(all_non_pending_pods_scheduled(s.pods, res)));
      double price = 0;
      for (auto& cnd : cands) price += cnd.price;
      v.set("candidatePrice", oj::Value::number(price));
      out.set("verdict", v);
    }
    return out;
  }
}

// Unit-level algebra probes used by the golden-vector tests.
//   {"fn":"intersection","a":req,"b":req}            -> requirement
//   {"fn":"has_intersection","a":req,"b":req}        -> bool
//   {"fn":"has","a":req,"value":"x"}                 -> bool
//   {"fn":"describe","a":req}                        -> requirement (operator, len)
//   {"fn":"compatible","a":[req],"b":[req],"allowUndefinedWellKnown":bool} -> bool   (a.Compatible(b, ...))
//   {"fn":"intersects","a":[req],"b":[req]}          -> bool
//   {"fn":"resources","op":"max|min|merge|subtract|fits","lists":[{..},..]} -> resource list / bool
//   {"fn":"tolerates","taints":[..],"tolerations":[..]} -> bool
//   {"fn":"sort_by_key","keys":[ints]}               -> permutation produced by Go's sort.Slice on (index,key) pairs
//   {"fn":"quantity","value":"1.8G"}                 -> nano-unit decimal string
char* oracle_eval_json(const char* query_json) {
  try {
    oj::Value q = oj::Parser(query_json).parse();
    auto& reg = labels_registry();
    reg = Labels();
    for (auto& k : q.at("wellKnownLabels").items()) reg.well_known.insert(sym(k.s()));
    std::string fn = q.at("fn").s();
    oj::Value out = oj::Value::object();
    if (fn == "intersection") {
      out.set("result", req_to_json(req_from_json(q.at("a")).intersection(req_from_json(q.at("b")))));
    } else if (fn == "has_intersection") {
      out.set("result", oj::Value::boolean(req_from_json(q.at("a")).has_intersection(req_from_json(q.at("b")))));
    } else if (fn == "has") {
      out.set("result", oj::Value::boolean(req_from_json(q.at("a")).has(sym(q.at("value").s()))));
    } else if (fn == "describe") {
      Requirement r = req_from_json(q.at("a"));
      oj::Value d = req_to_json(r);
      d.set("len", oj::Value::integer(r.len()));
      out.set("result", d);
    } else if (fn == "compatible") {
      std::string why;
      bool ok = reqs_from_json(q.at("a")).compatible(reqs_from_json(q.at("b")), q.at("allowUndefinedWellKnown").boolean_or(false), &why);
      out.set("result", oj::Value::boolean(ok));
      out.set("why", oj::Value::string(why));
    } else if (fn == "intersects") {
      out.set("result", oj::Value::boolean(reqs_from_json(q.at("a")).intersects(reqs_from_json(q.at("b")))));
    } else if (fn == "add") {
      Requirements r = reqs_from_json(q.at("a"));
      r.add_all(reqs_from_json(q.at("b")));
      out.set("result", reqs_to_json(r));
    } else if (fn == "resources") {
      std::vector<ResourceList> lists;
      for (auto& l : q.at("lists").items()) lists.push_back(parse_resources(l));
      std::vector<const ResourceList*> ptrs;
      for (auto& l : lists) ptrs.push_back(&l);
      std::string op = q.at("op").s();
      if (op == "max") out.set("result", res_to_json(res_max(ptrs)));
      else if (op == "min") out.set("result", res_to_json(res_min(ptrs)));
      else if (op == "merge") { ResourceList r; for (auto& l : lists) r = res_merge(r, l); out.set("result", res_to_json(r)); }
      else if (op == "subtract") out.set("result", res_to_json(res_subtract(lists.at(0), lists.at(1))));
      else if (op == "fits") out.set("result", oj::Value::boolean(res_fits(lists.at(0), lists.at(1))));
      else throw std::runtime_error("bad resources op");
    } else if (fn == "tolerates") {
      std::vector<Toleration> tols;
      for (auto& t : q.at("tolerations").items()) tols.push_back({sym(t.at("key").s()), sym(t.at("operator").s()), sym(t.at("value").s()), sym(t.at("effect").s())});
      out.set("result", oj::Value::boolean(taints_tolerated(parse_taints(q.at("taints")), tols)));
    } else if (fn == "sort_by_key") {
      std::vector<std::pair<long long, int>> v;
      int i = 0;
      for (auto& k : q.at("keys").items()) v.push_back({k.i(), i++});
      go_sort_slice(v, [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first < b.first; });
      oj::Value perm = oj::Value::array();
      for (auto& e : v) perm.push(oj::Value::integer(e.second));
      out.set("result", perm);
    } else if (fn == "order_trace") {
      // ops: -1 = append a new claim (count 1), i >= 0 = add one pod to claim id i. Before every op the claims are
      // re-sorted with Go's sort.Slice on the pod count, exactly like scheduler.go:598. Returns the final id order.
      std::vector<std::pair<long long, int>> v;
      for (auto& o : q.at("ops").items()) {
        go_sort_slice(v, [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first < b.first; });
        long long op = o.i();
        if (op < 0) v.push_back({1, (int)v.size()});
        else for (auto& e : v) if (e.second == (int)op) { e.first++; break; }
      }
      go_sort_slice(v, [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first < b.first; });
      oj::Value perm = oj::Value::array();
      for (auto& e : v) perm.push(oj::Value::integer(e.second));
      out.set("result", perm);
    } else if (fn == "quantity") {
      out.set("result", oj::Value::string(i128_to_string(parse_quantity(q.at("value").s()))));
    } else {
      throw std::runtime_error("unknown fn " + fn);
    }
    return dup_out(out);
  } catch (const std::exception& e) {
    return err_out(e.what());
  }
}

}  // extern "C"

#ifdef ORACLE_MAIN
#include <fstream>
#include <iostream>
#include <sstream>
int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: oracle_cli problem.json [eval]\n"); return 2; }
  std::ifstream f(argv[1]);
  std::stringstream ss;
  ss << f.rdbuf();
  char* out = (argc > 2 && std::string(argv[2]) == "eval") ? oracle_eval_json(ss.str().c_str()) : oracle_solve_json(ss.str().c_str());
  puts(out);
  oracle_free(out);
  return 0;
}
#endif
