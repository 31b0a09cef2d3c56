// ksched.cpp — host side of the solver above the C ABI: the part of NewScheduler/Solve that is object wrangling rather
// than compute. It mirrors what the Go shim (go/ksolve_shim.go) does inside Karpenter:
//   1. NewScheduler inputs (provisioner.go:265-360): NodePools ordered by weight (nodepool.go:161-171), one
//      NodeClaimTemplate per pool (nodeclaimtemplate.go:66-94), instance types, pods;
//   2. flatten them into the KSP arrays of include/ksolve.h (dictionary-encode labels, exact integer resources,
//      PodData requirement sets — requirements.go:74-118, scheduler.go:554-580 —, toleration masks, the
//      Preferences.Relax ladder — preferences.go:38-57);
//   3. ksolve_create / ksolve_solve on the device library;
//   4. rehydrate Results (scheduler.go:281-286) from the flat results.
// Input and output are JSON documents (schema: karpenter_amd/fixtures.py) so that Python tests can read like the
// reference's Go tests. There is NO scheduling logic here: requirement intersection reuses the same flat algebra the
// kernels use (csrc/reqalg.h), and anything the device build cannot solve is reported as "unsupported", never solved
// on the CPU.
#include <dlfcn.h>
#include <pthread.h>

#include <algorithm>
#include <climits>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstring>
#include <functional>
#include <map>
#include <unordered_map>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ksolve.h"
#include "../csrc/reqalg.h"
#include "json_mini.hpp"

namespace ks {
// ReqBuf::ref() hides minv when no key has minValues; the flattener always wants it.
inline ReqRef reqbuf_ref_with_minv(const ReqBuf& b) { ReqRef r = b.ref(); r.minv = b.minv; return r; }
}  // namespace ks

namespace {

typedef __int128 i128;
using kj::Value;

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- resource.Quantity -> exact int128 nano-units ----
// resource.Quantity saturates at +-(2^63 - 1) in its own scale; anything that would not fit 2^100 nano-units is far beyond
// every real request and is refused loudly instead of wrapping to a small number (a pod asking for cpu "1e400000000" must
// not be packed as if it asked for nothing).
static const i128 kQuantityMax = (i128)1 << 100;
bool go_atoi(const std::string& s, long long& out);
i128 parse_quantity(const std::string& s) {
  if (s.empty()) throw std::runtime_error("quantity: empty");
  size_t i = 0;
  bool neg = false;
  if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; ++i; }
  i128 mant = 0;
  int frac = 0;
  bool dot = false, any = false;
  for (; i < s.size(); ++i) {
    char c = s[i];
    if (c >= '0' && c <= '9') {
      if (mant > kQuantityMax / 10) throw Unsupported("quantity out of range (resource.Quantity would saturate): " + s);
      mant = mant * 10 + (c - '0'); if (dot) ++frac; any = true;
    }
    else if (c == '.' && !dot) dot = true;
    else break;
  }
  if (!any) throw std::runtime_error("quantity: '" + s + "'");
  std::string suf = s.substr(i);
  static const std::map<std::string, int> dec = {{"", 0}, {"n", -9}, {"u", -6}, {"m", -3}, {"k", 3}, {"M", 6}, {"G", 9}, {"T", 12}, {"P", 15}, {"E", 18}};
  static const std::map<std::string, int> bin = {{"Ki", 10}, {"Mi", 20}, {"Gi", 30}, {"Ti", 40}, {"Pi", 50}, {"Ei", 60}};
  int e = 0;
  i128 v = mant;
  auto d = dec.find(suf);
  if (d != dec.end()) e = d->second;
  else if (bin.count(suf)) {
    if (v > (kQuantityMax >> bin.at(suf))) throw Unsupported("quantity out of range (resource.Quantity would saturate): " + s);
    v *= (i128)1 << bin.at(suf);
  } else if (!suf.empty() && (suf[0] == 'e' || suf[0] == 'E')) {
    long long ee = 0;
    if (!go_atoi(suf.substr(1), ee) || ee > 40 || ee < -40) throw Unsupported("quantity exponent out of range: " + s);   // never a 2^31-step loop, never a silent wrap
    e = (int)ee;
  } else throw std::runtime_error("quantity suffix: '" + s + "'");
  e += 9 - frac;
  for (; e > 0; --e) {
    if (v > kQuantityMax / 10) throw Unsupported("quantity out of range (resource.Quantity would saturate): " + s);
    v *= 10;
  }
  for (; e < 0; ++e) { if (v % 10) throw Unsupported("quantity finer than nano: " + s); v /= 10; }
  return neg ? -v : v;
}
std::string i128_str(i128 v) {
  if (v == 0) return "0";
  bool neg = v < 0;
  unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
  std::string s;
  while (u) { s += (char)('0' + (int)(u % 10)); u /= 10; }
  if (neg) s += '-';
  std::reverse(s.begin(), s.end());
  return s;
}
bool go_atoi(const std::string& s, long long& out) {  // strconv.Atoi
  if (s.empty()) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; }
  if (i == s.size()) return false;
  unsigned long long v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    unsigned d = s[i] - '0';
    if (v > (ULLONG_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  if (!neg && v > (unsigned long long)LLONG_MAX) return false;
  if (neg && v > (unsigned long long)LLONG_MAX + 1ULL) return false;
  out = neg ? (long long)(0 - v) : (long long)v;
  return true;
}

const char* kHostname = "kubernetes.io/hostname";
const char* kZone = "topology.kubernetes.io/zone";
const char* kInstanceType = "node.kubernetes.io/instance-type";
const char* kCapacityType = "karpenter.sh/capacity-type";
const char* kNodePool = "karpenter.sh/nodepool";

std::string normalize_key(const std::string& k) {  // v1.NormalizedLabels — labels.go:121-127
  static const std::map<std::string, std::string> n = {{"failure-domain.beta.kubernetes.io/zone", kZone},
                                                       {"beta.kubernetes.io/arch", "kubernetes.io/arch"},
                                                       {"beta.kubernetes.io/os", "kubernetes.io/os"},
                                                       {"beta.kubernetes.io/instance-type", kInstanceType},
                                                       {"failure-domain.beta.kubernetes.io/region", "topology.kubernetes.io/region"}};
  auto it = n.find(k);
  return it == n.end() ? k : it->second;
}

struct Expr { std::string key, op; std::vector<std::string> values; int min_values = -1; };
Expr parse_expr(const Value& v) {
  Expr e;
  e.key = normalize_key(v.at("key").s());
  e.op = v.at("operator").s();
  for (auto& x : v.at("values").items()) e.values.push_back(x.s());
  if (v.has("minValues") && !v.at("minValues").is_null()) e.min_values = (int)v.at("minValues").i();
  return e;
}
std::vector<Expr> parse_exprs(const Value& v) { std::vector<Expr> out; for (auto& e : v.items()) out.push_back(parse_expr(e)); return out; }
std::vector<Expr> label_exprs(const Value& labels) {
  std::vector<Expr> out;
  for (auto& kv : labels.members()) out.push_back(Expr{normalize_key(kv.first), "In", {kv.second.s()}, -1});
  return out;
}

struct Taint { std::string key, value, effect; bool operator<(const Taint& o) const { return std::tie(key, value, effect) < std::tie(o.key, o.value, o.effect); } };
struct Toleration { std::string key, op, value, effect; };
// corev1.Toleration.ToleratesTaint(logger, taint, enableComparisonOperators=true) — called at taints.go:89
bool tolerates(const Toleration& t, const Taint& x) {
  if (!t.effect.empty() && t.effect != x.effect) return false;
  if (!t.key.empty() && t.key != x.key) return false;
  if (t.op.empty() || t.op == "Equal") return t.value == x.value;
  if (t.op == "Exists") return true;
  if (t.op == "Lt" || t.op == "Gt") {
    long long tv, xv;
    if (!go_atoi(t.value, tv) || !go_atoi(x.value, xv)) return false;
    return t.op == "Lt" ? xv < tv : xv > tv;
  }
  return false;
}

// ---- dictionaries ----
struct Dictionary {
  std::vector<std::string> keys;
  std::map<std::string, int> key_index;
  std::vector<std::vector<std::string>> values;             // per key
  std::vector<std::map<std::string, int>> value_index;
  int key(const std::string& k) {
    auto it = key_index.find(k);
    if (it != key_index.end()) return it->second;
    int i = (int)keys.size();
    keys.push_back(k); key_index[k] = i; values.emplace_back(); value_index.emplace_back();
    return i;
  }
  int value(int k, const std::string& v) {
    auto it = value_index[k].find(v);
    if (it != value_index[k].end()) return it->second;
    int i = (int)values[k].size();
    values[k].push_back(v); value_index[k][v] = i;
    return i;
  }
  void note(const Expr& e) { int k = key(e.key); if (e.op == "In" || e.op == "NotIn") for (auto& v : e.values) value(k, v); }
};

// A table of requirement sets under construction (host memory, ABI layout).
// fn(i) for i in [0, n) on a few threads, contiguous ranges (the probes of a sweep are independent — descriptors, verdicts — and so
// are the rows of a million pods)
// The worker threads are kept (round 6): a 10,000-probe sweep calls this twice — descriptors, verdicts — for a few hundred microseconds
// of work each, and creating and joining eight threads cost as much as the work. One pool per process, created on first use and never
// destroyed (its threads sleep on a condition variable; the process's exit ends them); a caller that finds the pool busy — another thread
// of the process is inside a parallel_for — starts threads of its own as before.
struct WorkerPool {
  std::mutex busy;                                   // one parallel_for at a time
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  std::vector<std::thread> workers;
  const std::function<void(size_t)>* job = nullptr;  // job(t) for t in 1..nt-1 (the caller runs t = 0)
  size_t nt = 0, generation = 0, pending = 0;
  explicit WorkerPool(size_t n_workers) {
    for (size_t w = 0; w < n_workers; ++w) workers.emplace_back([this, w]() {
      size_t seen = 0;
      for (;;) {
        const std::function<void(size_t)>* j = nullptr;
        {
          std::unique_lock<std::mutex> lk(m);
          cv_go.wait(lk, [&] { return generation != seen; });
          seen = generation;
          if (w + 1 < nt) j = job;
        }
        if (j) {
          (*j)(w + 1);
          std::lock_guard<std::mutex> lk(m);
          if (--pending == 0) cv_done.notify_one();
        }
      }
    });
    for (auto& t : workers) t.detach();
  }
  void run(size_t n_threads, const std::function<void(size_t)>& f) {   // f(t) for t in [0, n_threads), n_threads <= workers + 1
    {
      std::lock_guard<std::mutex> lk(m);
      job = &f; nt = n_threads; pending = n_threads - 1; generation++;
    }
    cv_go.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
    job = nullptr;
  }
};
static WorkerPool* g_worker_pool = nullptr;           // leaked on purpose: no destructor runs against sleeping threads at exit
static std::mutex g_worker_pool_m;
static WorkerPool* worker_pool() {
  std::lock_guard<std::mutex> lk(g_worker_pool_m);
  if (!g_worker_pool) {
    static bool fork_hook = false;
    if (!fork_hook) { fork_hook = true; pthread_atfork(nullptr, nullptr, [] { g_worker_pool = nullptr; }); }   // a forked child has none of the threads: it makes its own pool
    g_worker_pool = new WorkerPool(7);
  }
  return g_worker_pool;
}
template <class F>
static void parallel_for(size_t n, F fn) {
  const size_t nt = n < 1024 ? 1 : std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency()));
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  std::vector<std::string> errors(nt);
  auto part = [&](size_t t) {
    try { for (size_t i = n * t / nt, e = n * (t + 1) / nt; i < e; ++i) fn(i); } catch (const std::exception& e) { errors[t] = e.what(); }
  };
  WorkerPool* wp = worker_pool();
  if (wp->busy.try_lock()) {
    std::lock_guard<std::mutex> hold(wp->busy, std::adopt_lock);
    const std::function<void(size_t)> f = part;
    wp->run(nt, f);
  } else {
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nt; ++t) pool.emplace_back(part, t);
    for (auto& th : pool) th.join();
  }
  for (auto& e : errors) if (!e.empty()) throw std::runtime_error(e);
}

// UID texts of the explicit pods of a problem (they come first); the pods of a group have none — theirs are derived from
// (seed, position) on demand — so a million group pods do not cost a million empty strings here and again in the session.
struct UidTexts {
  std::vector<std::string> explicit_;
  const std::string& operator[](size_t p) const { static const std::string none; return p < explicit_.size() ? explicit_[p] : none; }
  void push_back(const std::string& u) { explicit_.push_back(u); }
};

struct ReqTableBuilder {
  int n = 0, req_words = 0, n_keys = 0;
  std::vector<uint64_t> mask;
  std::vector<uint32_t> defined, complement, has_gte, has_lte;
  std::vector<int64_t> gte, lte;
  std::vector<int32_t> minv;
  // the bounds and minValues columns (n * n_keys each) exist once a set carries one: a table of a million pod rows without
  // Gt / Lt / minValues hands the ABI null columns (absent == none) instead of 40 bytes per row and key of zeroes
  void init(int n_, int rw, int nk, bool lazy_columns = false) {   // lazy_columns: the pod-row tables
    n = n_; req_words = rw; n_keys = nk;
    mask.assign((size_t)n * rw, 0); defined.assign(n, 0); complement.assign(n, 0); has_gte.assign(n, 0); has_lte.assign(n, 0);
    gte.clear(); lte.clear(); minv.clear();
    if (!lazy_columns) { gte.assign((size_t)n * nk, 0); lte.assign((size_t)n * nk, 0); minv.assign((size_t)n * nk, -1); }
  }
  // the lazily created columns up front (put() from several threads must not be the one that creates them)
  void ensure_columns(bool bounds, bool min_values) {
    if (bounds && gte.empty()) { gte.assign((size_t)n * n_keys, 0); lte.assign((size_t)n * n_keys, 0); }
    if (min_values && minv.empty()) minv.assign((size_t)n * n_keys, -1);
  }
  void put(int e, const ks::ReqBuf& b) {
    for (int w = 0; w < req_words; ++w) mask[(size_t)e * req_words + w] = b.mask[w];
    defined[e] = b.defined; complement[e] = b.complement; has_gte[e] = b.has_gte; has_lte[e] = b.has_lte;
    if ((b.has_gte | b.has_lte) && gte.empty()) { gte.assign((size_t)n * n_keys, 0); lte.assign((size_t)n * n_keys, 0); }
    if (b.has_minv && minv.empty()) minv.assign((size_t)n * n_keys, -1);
    if (!gte.empty()) for (int k = 0; k < n_keys; ++k) { gte[(size_t)e * n_keys + k] = b.gte[k]; lte[(size_t)e * n_keys + k] = b.lte[k]; }
    if (!minv.empty()) for (int k = 0; k < n_keys; ++k) minv[(size_t)e * n_keys + k] = b.minv[k];
  }
  ksolve_reqsets view() const {
    ksolve_reqsets r{};
    r.n = (uint32_t)n; r.mask = mask.data(); r.defined = defined.data(); r.complement = complement.data();
    r.has_gte = has_gte.data(); r.has_lte = has_lte.data();
    r.gte = gte.empty() ? nullptr : gte.data(); r.lte = lte.empty() ? nullptr : lte.data(); r.min_values = minv.empty() ? nullptr : minv.data();
    return r;
  }
};

struct Flattener {
  Dictionary dict;
  ks::Dict kd{};
  std::vector<uint32_t> key_word_off;
  std::vector<int64_t> value_int;
  std::vector<uint64_t> value_is_int, value_valid;
  std::set<std::string> well_known;

  void finalize_dictionary(int n_its_words_min) {
    int nk = (int)dict.keys.size();
    if (nk > KSOLVE_MAX_KEYS) throw Unsupported("more than 32 distinct requirement keys");
    key_word_off.assign(nk + 1, 0);
    for (int k = 0; k < nk; ++k) {
      int words = std::max(1, ((int)dict.values[k].size() + 63) / 64);
      if (dict.keys[k] == kInstanceType) words = std::max(words, n_its_words_min);
      key_word_off[k + 1] = key_word_off[k] + words;
    }
    int rw = key_word_off[nk];
    if (rw > ks::kMaxReqWords) throw Unsupported("requirement dictionaries need more than 96 mask words");
    value_int.assign((size_t)rw * 64, 0); value_is_int.assign(rw, 0); value_valid.assign(rw, 0);
    for (int k = 0; k < nk; ++k)
      for (size_t v = 0; v < dict.values[k].size(); ++v) {
        size_t bitpos = (size_t)key_word_off[k] * 64 + v;
        value_valid[bitpos / 64] |= 1ull << (bitpos % 64);
        long long iv;
        if (go_atoi(dict.values[k][v], iv)) { value_int[bitpos] = iv; value_is_int[bitpos / 64] |= 1ull << (bitpos % 64); }
      }
    kd.n_keys = nk; kd.req_words = rw;
    for (int k = 0; k <= nk; ++k) kd.key_word_off[k] = key_word_off[k];
    kd.well_known_mask = 0;
    for (int k = 0; k < nk; ++k) if (well_known.count(dict.keys[k])) kd.well_known_mask |= 1u << k;
    auto find = [&](const char* name) { auto it = dict.key_index.find(name); return it == dict.key_index.end() ? -1 : it->second; };
    kd.key_it = find(kInstanceType); kd.key_zone = find(kZone); kd.key_ct = find(kCapacityType); kd.key_hostname = find(kHostname);
    kd.value_int = value_int.data(); kd.value_is_int = value_is_int.data(); kd.value_valid = value_valid.data();
  }
  // NewRequirementWithFlexibility (requirement.go:48-110) in flat form
  void encode(const Expr& e, ks::ReqBuf& b) {
    memset(&b, 0, sizeof(b));
    for (int k = 0; k < ks::kMaxKeys; ++k) b.minv[k] = -1;
    int k = dict.key_index.at(e.key);
    uint32_t kb = 1u << k;
    b.defined = kb;
    auto set_vals = [&]() { for (auto& v : e.values) { size_t pos = (size_t)key_word_off[k] * 64 + dict.value_index[k].at(v); b.mask[pos / 64] |= 1ull << (pos % 64); } };
    auto atoi0 = [&](const std::string& s) { long long v = 0; go_atoi(s, v); return v; };
    if (e.min_values >= 0) { b.minv[k] = e.min_values; b.has_minv = kb; }
    if (e.op == "In") { set_vals(); return; }
    if (e.op == "DoesNotExist") return;
    b.complement = kb;
    if (e.op == "NotIn") { set_vals(); return; }
    if (e.op == "Exists") return;
    long long v = e.values.empty() ? 0 : atoi0(e.values[0]);
    if (e.op == "Gt") {
      if (v == LLONG_MAX) { b.complement = 0; b.minv[k] = -1; b.has_minv = 0; return; }  // Gt MaxInt matches nothing (requirement.go:85-88)
      b.has_gte = kb; b.gte[k] = v + 1;
    } else if (e.op == "Lt") { b.has_lte = kb; b.lte[k] = v - 1; }
    else if (e.op == "Gte") { b.has_gte = kb; b.gte[k] = v; }
    else if (e.op == "Lte") { b.has_lte = kb; b.lte[k] = v; }
    else throw std::runtime_error("bad operator " + e.op);
  }
  static void clear(ks::ReqBuf& b) { memset(&b, 0, sizeof(b)); for (int k = 0; k < ks::kMaxKeys; ++k) b.minv[k] = -1; }
};

// metav1.LabelSelector as labels.Selector: nil matches nothing, empty matches everything (topologygroup.go:101-104,:443)
struct SelExpr {
  std::string key, op;
  std::set<std::string> values;
  bool operator<(const SelExpr& o) const { return std::tie(key, op, values) < std::tie(o.key, o.op, o.values); }
};
struct Selector {
  bool nil = true;
  std::map<std::string, std::string> match_labels;
  std::vector<SelExpr> exprs;
  bool valid() const {
    for (auto& e : exprs) {
      if (e.op == "In" || e.op == "NotIn") { if (e.values.empty()) return false; }
      else if (e.op == "Exists" || e.op == "DoesNotExist") { if (!e.values.empty()) return false; }
      else return false;
    }
    return true;
  }
  bool matches(const std::map<std::string, std::string>& labels) const {
    if (nil || !valid()) return false;
    for (auto& kv : match_labels) { auto it = labels.find(kv.first); if (it == labels.end() || it->second != kv.second) return false; }
    for (auto& e : exprs) {
      auto it = labels.find(e.key);
      if (e.op == "In") { if (it == labels.end() || !e.values.count(it->second)) return false; }
      else if (e.op == "NotIn") { if (it != labels.end() && e.values.count(it->second)) return false; }
      else if (e.op == "Exists") { if (it == labels.end()) return false; }
      else if (e.op == "DoesNotExist") { if (it != labels.end()) return false; }
    }
    return true;
  }
  std::string canon() const {   // what TopologyGroup.Hash() sees of the selector (slices hashed as sets)
    std::string out = nil ? "nil;" : "sel;";
    for (auto& kv : match_labels) out += kv.first + "=" + kv.second + ",";
    out += ";";
    std::set<SelExpr> ex(exprs.begin(), exprs.end());
    for (auto& e : ex) { out += e.key + " " + e.op + " ["; for (auto& v : e.values) out += v + ","; out += "];"; }
    return out;
  }
};
struct Tsc {
  int max_skew = 1, min_domains = -1;
  std::string key, when = "DoNotSchedule", taint_policy, affinity_policy;   // policies: "" = nil
  Selector sel;
};
// `namespaces` is already buildNamespaceList's answer when `resolved` (the term had a namespaceSelector): it may then be
// empty, which selects nothing, unlike an absent list, which means the pod's own namespace (topology.go:536-557)
struct AffTerm { Selector sel; std::string key; std::vector<std::string> namespaces; bool resolved = false; };

// ---- pod model (only what the flattener needs) ----
// scheduling.HostPort (hostportusage.go:39-62); the IP in the canonical text net.ParseIP would print ("<nil>" when it
// does not parse), hostIP "" read as 0.0.0.0 (hostportusage.go:103-106)
struct HostPort {
  std::string ip; int port; std::string protocol;
  bool unspecified() const { return ip == "0.0.0.0" || ip == "::"; }
  bool matches(const HostPort& o) const { return protocol == o.protocol && port == o.port && (ip == o.ip || unspecified() || o.unspecified()); }
  bool operator==(const HostPort& o) const { return ip == o.ip && port == o.port && protocol == o.protocol; }
};
static std::string canonical_ip(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  if (s.rfind("::ffff:", 0) == 0 && s.find('.') != std::string::npos) s = s.substr(7);
  if (s.find(':') != std::string::npos) {
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
static std::vector<HostPort> parse_host_ports(const Value& v) {
  std::vector<HostPort> out;
  if (v.is_null()) return out;
  for (auto& e : v.items()) {
    const int port = (int)e.at("port").i(0);
    if (port == 0) continue;
    const std::string ip = e.at("ip").s("");
    out.push_back({canonical_ip(ip.empty() ? "0.0.0.0" : ip), port, e.at("protocol").s("TCP")});
  }
  return out;
}

struct PodSpec {
  std::vector<HostPort> host_ports;   // GetHostPorts (hostportusage.go:93-117)
  std::string ns = "default", phase = "Pending";
  std::map<std::string, std::string> labels;
  std::vector<Tsc> tscs;
  bool has_pod_affinity = false, has_pod_anti = false;
  std::vector<AffTerm> aff_required, anti_required;
  std::vector<std::pair<int, AffTerm>> aff_preferred, anti_preferred;
  std::string uid;
  long long creation = 0;
  bool pending = true;
  std::string node_name;
  std::map<std::string, i128> requests;
  Value node_selector;
  bool has_node_affinity = false, has_required = false;
  std::vector<std::vector<Expr>> required_terms;
  std::vector<std::pair<int, std::vector<Expr>>> preferred;  // (weight, exprs)
  std::vector<Toleration> tolerations;
  std::vector<std::vector<Expr>> volume_requirements;        // volumeReqsByPod[uid] (scheduler.go:138, :572): alternatives, in order
  std::vector<std::pair<std::string, std::string>> volumes;   // scheduling.GetVolumes(pod) (volumeusage.go:83-114): <CSI driver, PVC>
};

// Go's insertion sort (sort.Slice on <= 12 elements is a stable insertion sort; pods with more than 12 preferred
// node-affinity terms are rejected so the unstable pdqsort path is never needed here) — requirements.go:102
template <class T, class L>
void small_stable_sort(std::vector<T>& v, L less) {
  if (v.size() > 12) throw Unsupported("more than 12 preferred node affinity terms");
  std::stable_sort(v.begin(), v.end(), less);
}

struct Api {
  void* lib = nullptr;
  decltype(&ksolve_create) create = nullptr;
  decltype(&ksolve_solve) solve = nullptr;
  decltype(&ksolve_results_free) results_free = nullptr;
  decltype(&ksolve_destroy) destroy = nullptr;
  decltype(&ksolve_last_error) last_error = nullptr;
  decltype(&ksolve_last_kernel_ms) kernel_ms = nullptr;
  decltype(&ksolve_cancel) cancel = nullptr;
  bool load(const char* path, std::string& err) {
    lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { err = std::string("cannot load solver library: ") + dlerror(); return false; }
    create = (decltype(create))dlsym(lib, "ksolve_create");
    solve = (decltype(solve))dlsym(lib, "ksolve_solve");
    results_free = (decltype(results_free))dlsym(lib, "ksolve_results_free");
    destroy = (decltype(destroy))dlsym(lib, "ksolve_destroy");
    last_error = (decltype(last_error))dlsym(lib, "ksolve_last_error");
    kernel_ms = (decltype(kernel_ms))dlsym(lib, "ksolve_last_kernel_ms");
    cancel = (decltype(cancel))dlsym(lib, "ksolve_cancel");
    if (!create || !solve || !results_free || !destroy || !last_error) { err = "solver library lacks ksolve_* symbols"; return false; }
    return true;
  }
};

std::map<std::string, i128> parse_resources(const Value& v) {
  std::map<std::string, i128> r;
  for (auto& kv : v.members()) {
    if (kv.second.kind == Value::Str) r[kv.first] = parse_quantity(kv.second.str);
    else if (kv.second.kind == Value::Num && kv.second.is_int) r[kv.first] = (i128)kv.second.inum * 1000000000;
    else throw std::runtime_error("resource quantity must be a string or an integer");
  }
  return r;
}

uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// uid of pod i of a podGroup (fixtures.py): 128 bits from splitmix64, printed as a UUID
void group_uid(uint64_t seed, uint64_t i, uint64_t& hi, uint64_t& lo, std::string* text) {
  uint64_t st = seed * 0x9E3779B97F4A7C15ULL + i * 0xD1B54A32D192ED03ULL + 0x2545F4914F6CDD1DULL;
  uint64_t a = splitmix64(st), b = splitmix64(st);
  hi = a; lo = b;
  if (text) {
    char buf[40];
    snprintf(buf, sizeof buf, "%08x-%04x-%04x-%04x-%012llx", (unsigned)(a >> 32), (unsigned)((a >> 16) & 0xffff), (unsigned)(a & 0xffff),
             (unsigned)(b >> 48), (unsigned long long)(b & 0xffffffffffffULL));
    *text = buf;
  }
}
bool parse_uuid(const std::string& s, uint64_t& hi, uint64_t& lo) {
  if (s.size() != 36) return false;
  uint64_t v[2] = {0, 0};
  int nib = 0;
  for (size_t i = 0; i < 36; ++i) {
    char c = s[i];
    if (i == 8 || i == 13 || i == 18 || i == 23) { if (c != '-') return false; continue; }
    int d;
    if (c >= '0' && c <= '9') d = c - '0';
    else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
    else return false;  // upper-case hex sorts differently as a string: use the rank fallback
    v[nib / 16] = (v[nib / 16] << 4) | (unsigned)d;
    nib++;
  }
  hi = v[0]; lo = v[1];
  return true;
}

}  // namespace


// ---------------------------------------------------------------------------------------------------------------
static std::vector<int32_t> b64_int32(const std::string& t) {
  std::vector<uint8_t> bytes;
  uint32_t acc = 0; int bits = 0;
  for (char c : t) {
    int v = (c >= 'A' && c <= 'Z') ? c - 'A' : (c >= 'a' && c <= 'z') ? c - 'a' + 26 : (c >= '0' && c <= '9') ? c - '0' + 52 : c == '+' ? 62 : c == '/' ? 63 : -1;
    if (v < 0) continue;   // padding / whitespace
    acc = (acc << 6) | (uint32_t)v; bits += 6;
    if (bits >= 8) { bits -= 8; bytes.push_back((uint8_t)(acc >> bits)); }
  }
  std::vector<int32_t> out(bytes.size() / 4);
  if (!out.empty()) memcpy(out.data(), bytes.data(), out.size() * 4);
  return out;
}
static Selector parse_selector(const Value& v) {
  Selector s;
  if (v.is_null()) return s;
  s.nil = false;
  for (auto& kv : v.at("matchLabels").members()) s.match_labels[kv.first] = kv.second.s();
  for (auto& e : v.at("matchExpressions").items()) {
    SelExpr x;
    x.key = e.at("key").s(); x.op = e.at("operator").s();
    for (auto& val : e.at("values").items()) x.values.insert(val.s());
    s.exprs.push_back(x);
  }
  return s;
}
// The namespace lister of the problem being flattened (root["namespaces"]: name + labels), standing in for the kube
// client that buildNamespaceList queries (topology.go:548-550).
static thread_local const std::vector<std::pair<std::string, std::map<std::string, std::string>>>* t_namespace_lister = nullptr;
static AffTerm parse_aff_term(const Value& v) {
  AffTerm t;
  t.sel = parse_selector(v.at("labelSelector"));
  t.key = v.at("topologyKey").s();
  for (auto& n : v.at("namespaces").items()) t.namespaces.push_back(n.s());
  if (v.has("namespaceSelector") && !v.at("namespaceSelector").is_null()) {
    Selector ns = parse_selector(v.at("namespaceSelector"));
    if (!ns.valid()) throw std::runtime_error("parsing selector: invalid namespaceSelector");   // topology.go:545-547
    std::set<std::string> out(t.namespaces.begin(), t.namespaces.end());
    if (t_namespace_lister) for (auto& n : *t_namespace_lister) if (ns.matches(n.second)) out.insert(n.first);
    t.namespaces.assign(out.begin(), out.end());
    t.resolved = true;
  }
  return t;
}
static PodSpec parse_pod(const Value& v) {
  PodSpec p;
  p.host_ports = parse_host_ports(v.at("hostPorts"));
  p.uid = v.at("uid").s();
  p.creation = v.at("creationTimestamp").i(0);
  p.pending = v.at("phase").s("Pending") == "Pending";
  p.node_name = v.at("nodeName").s("");
  p.requests = parse_resources(v.at("requests"));
  p.node_selector = v.at("nodeSelector");
  for (auto& alt : v.at("volumeRequirements").items()) p.volume_requirements.push_back(parse_exprs(alt));
  for (auto& vv : v.at("volumes").items()) p.volumes.push_back({vv.at("driver").s(), vv.at("pvc").s()});
  const Value& na = v.at("nodeAffinity");
  if (!na.is_null()) {
    p.has_node_affinity = true;
    if (na.has("required") && !na.at("required").is_null()) {
      p.has_required = true;
      for (auto& t : na.at("required").items()) p.required_terms.push_back(parse_exprs(t));
    }
    for (auto& t : na.at("preferred").items()) p.preferred.push_back({(int)t.at("weight").i(), parse_exprs(t.at("matchExpressions"))});
  }
  for (auto& t : v.at("tolerations").items()) p.tolerations.push_back({t.at("key").s(), t.at("operator").s(), t.at("value").s(), t.at("effect").s()});
  p.ns = v.at("namespace").s("default");
  p.phase = v.at("phase").s("Pending");
  for (auto& kv : v.at("labels").members()) p.labels[kv.first] = kv.second.s();
  for (auto& c : v.at("topologySpreadConstraints").items()) {
    Tsc t;
    t.max_skew = (int)c.at("maxSkew").i(1);
    t.key = c.at("topologyKey").s();
    t.when = c.at("whenUnsatisfiable").s("DoNotSchedule");
    t.sel = parse_selector(c.at("labelSelector"));
    if (c.has("minDomains") && !c.at("minDomains").is_null()) t.min_domains = (int)c.at("minDomains").i();
    if (c.has("nodeTaintsPolicy") && !c.at("nodeTaintsPolicy").is_null()) t.taint_policy = c.at("nodeTaintsPolicy").s();
    if (c.has("nodeAffinityPolicy") && !c.at("nodeAffinityPolicy").is_null()) t.affinity_policy = c.at("nodeAffinityPolicy").s();
    // matchLabelKeys are merged into the selector (topology.go:470-478)
    for (auto& k : c.at("matchLabelKeys").items()) {
      auto it = p.labels.find(k.s());
      if (it != p.labels.end()) { t.sel.nil = false; t.sel.exprs.push_back(SelExpr{k.s(), "In", {it->second}}); }
    }
    p.tscs.push_back(t);
  }
  const Value& pa = v.at("podAffinity");
  if (!pa.is_null()) {
    p.has_pod_affinity = true;
    for (auto& t : pa.at("required").items()) p.aff_required.push_back(parse_aff_term(t));
    for (auto& t : pa.at("preferred").items()) p.aff_preferred.push_back({(int)t.at("weight").i(), parse_aff_term(t.at("term"))});
  }
  const Value& paa = v.at("podAntiAffinity");
  if (!paa.is_null()) {
    p.has_pod_anti = true;
    for (auto& t : paa.at("required").items()) p.anti_required.push_back(parse_aff_term(t));
    for (auto& t : paa.at("preferred").items()) p.anti_preferred.push_back({(int)t.at("weight").i(), parse_aff_term(t.at("term"))});
  }
  return p;
}

static char* dup_json(const Value& v) {
  std::string s;
  kj::write(v, s);
  char* out = (char*)malloc(s.size() + 1);
  memcpy(out, s.c_str(), s.size() + 1);
  return out;
}
static char* error_json(const char* kind, const std::string& msg) {
  Value o = Value::object();
  o.set("error", Value::string(msg));
  o.set("kind", Value::string(kind));
  return dup_json(o);
}

extern "C" void ksched_free(char* p) { free(p); }
// parse + write of one document (tests hold the library's JSON reader / writer against an independent implementation with it)
extern "C" char* ksched_json_roundtrip(const char* doc) {
  try {
    return dup_json(kj::Parser(doc).parse());
  } catch (const std::exception& e) {
    return error_json("parse", e.what());
  }
}
struct Session;
extern "C" uint32_t ksched_assignment(void* session, int32_t* assign, uint32_t* slot, uint32_t capacity);
extern "C" uint32_t ksched_pods_by_claim(void* session, uint32_t n_claims, uint32_t* claim_off, uint32_t* pods, uint32_t capacity);

// A problem flattened and resident on the device: what NewScheduler returns.
constexpr long long kMaxPodsPerProblem = 1ll << 24;

struct Session {
  Api api;
  ksolve_handle* handle = nullptr;
  Value root;
  Flattener fl;
  std::vector<std::string> pool_names, res_names, it_names, node_names;
  UidTexts uid_text;
  std::vector<uint8_t> node_initialized;
  std::vector<std::pair<uint64_t, uint64_t>> group_of_pod;
  std::vector<i128> scale;
  int n_pods = 0, n_rows = 0, n_its = 0, n_res = 0, it_words = 0, k_rid = -1, n_topo_groups = 0, n_alias_classes = 0;
  std::string error_kind, error;
  // probes of a resident cluster (ksched_probe): what a probe needs from the base session besides the device handle
  int n_templates = 0;
  std::vector<int> node_tmpl;              // template (NodePool) of each existing node, -1 = none / pool without limits
  std::vector<int64_t> node_limit_cap;     // [n_nodes][n_res+1] node capacity on the dimensions its pool limits (device units)
  std::vector<int64_t> tmpl_lim;           // [n_templates][n_res+1] remaining limits of the base problem
  std::unordered_map<std::string, int> node_index, pod_index;   // built on the first probe
  // sweeps of a resident cluster (ksched_sweep): which node every pod sits on, the verdict inputs computeConsolidation needs
  std::vector<int32_t> pod_node;           // existing node (sorted order) a pod is bound to, -1 = pending / on a deleting node: part of every simulation
  std::vector<uint8_t> pod_pending_flag, pod_deleting_flag;
  std::vector<int32_t> last_assign;      // want_results 2: the flat per-pod outputs of the last solve, for ksched_assignment
  std::vector<uint32_t> last_slot;
  std::vector<uint32_t> node_pod_off, node_pod_list, always_pods;   // CSR of the pods on each node; built on the first sweep
  std::vector<int32_t> node_input_index;   // position in the problem's stateNodes list -> sorted node index
  std::vector<int32_t> node_it;            // instance type of each existing node (its node.kubernetes.io/instance-type label), -1 = unknown
  std::vector<double> node_price;          // resolveNodePrice (disruption/types.go:113-127): the offering of its zone and capacity type, 0 = none
  std::vector<uint8_t> node_spot;          // capacity-type label == spot
  struct OffLite { int zone, ct, rid; double price; bool available; };
  std::vector<std::vector<OffLite>> it_offerings;
  std::vector<std::vector<Expr>> it_exprs;
  bool sweep_tables = false;
  bool strict_shared = false;              // no pod has a preference: PodData.StrictRequirements IS Requirements, one table on the device
  // a probe session: shares the base session's metadata, owns its handle
  Session* base = nullptr;
  std::vector<uint8_t> probe_member, probe_removed;
  int probe_pods = 0;
};

static char* session_error(Session* s) { char* r = error_json(s->error_kind.c_str(), s->error); return r; }

// NewScheduler: parse + flatten the problem document and upload it through ksolve_create. Returns a session handle
// (never null); ksched_error(session) is non-null when it failed.
extern "C" void* ksched_open(const char* problem_json, const char* solver_lib) {
  Session* S = new Session();
  Api& api = S->api;
  std::string err;
  if (!api.load(solver_lib, err)) { S->error_kind = "load"; S->error = err; return S; }
  ksolve_handle*& handle = S->handle;
  // KSCHED_TRACE=1: wall time of every phase of NewScheduler on stderr
  const bool tracing = getenv("KSCHED_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto trace = [&](const char* next) {
    if (!tracing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "ksched_open: %8.1f ms until '%s'\n", std::chrono::duration<double, std::milli>(now - t_last).count(), next);
    t_last = now;
  };
  try {
    S->root = kj::Parser(problem_json).parse();
    Value& root = S->root;
    Flattener& fl = S->fl;
    for (const char* k : {kNodePool, kZone, "topology.kubernetes.io/region", kInstanceType, "kubernetes.io/arch", "kubernetes.io/os", kCapacityType, "node.kubernetes.io/windows-build"}) fl.well_known.insert(k);
    for (auto& k : root.at("wellKnownLabels").items()) fl.well_known.insert(k.s());
    const Value& opts = root.at("options");
    bool ignore_prefs = opts.at("preferencePolicy").s("Respect") == "Ignore";

    trace("instance types");
    // ---- instance types ----
    const auto& its_json = root.at("instanceTypes").items();
    const int n_its = (int)its_json.size();
    std::map<std::string, int> it_index;
    Dictionary& D = fl.dict;
    int k_it = D.key(kInstanceType), k_zone = D.key(kZone), k_ct = D.key(kCapacityType);
    for (int i = 0; i < n_its; ++i) {
      const std::string name = its_json[i].at("name").s();
      if (it_index.count(name)) throw std::runtime_error("duplicate instance type " + name);
      it_index[name] = i;
      D.value(k_it, name);
    }
    // offering zones / capacity types first so their value indices are the offering cell coordinates
    std::vector<std::vector<Expr>> it_exprs(n_its);
    struct Off { int zone, ct; double price; bool available; int rid; std::map<std::string, i128> cap_ov; bool has_oh; std::map<std::string, i128> oh_ov; };
    std::vector<std::vector<Off>> it_offs(n_its);
    const char* kReservationID = "karpenter.sh/reservation-id";   // cloudprovider.ReservationIDLabel
    int k_rid = -1;
    std::vector<int32_t> resv_capacity;   // per reservation id: the most pessimistic ReservationCapacity (reservationmanager.go:45-60)
    for (int i = 0; i < n_its; ++i)
      for (auto& of : its_json[i].at("offerings").items()) {
        std::string zone, ct, rid;
        for (auto& e : of.at("requirements").items()) {
          Expr x = parse_expr(e);
          if (x.op != "In" || x.values.size() != 1) throw Unsupported("offering requirements must be single-valued In");
          if (x.key == kZone) zone = x.values[0];
          else if (x.key == kCapacityType) ct = x.values[0];
          else if (x.key == kReservationID) rid = x.values[0];
          else throw Unsupported("offering requirement on " + x.key);
        }
        if (zone.empty() || ct.empty()) throw std::runtime_error("offering without zone/capacity-type");
        // Offering.CapacityOverride / OverheadOverride (types.go:476-483): the offering belongs to an allocatable group of its own
        std::map<std::string, i128> cap_ov, oh_ov;
        bool has_oh = false;
        if (of.has("capacityOverride") && !of.at("capacityOverride").is_null()) cap_ov = parse_resources(of.at("capacityOverride"));
        if (of.has("overheadOverride") && !of.at("overheadOverride").is_null()) { has_oh = true; oh_ov = parse_resources(of.at("overheadOverride")); }
        if ((!cap_ov.empty() || has_oh) && ct == "reserved") throw Unsupported("capacity / overhead overrides on a reserved offering");
        int ridx = -1;
        if (ct == "reserved") {
          if (rid.empty()) throw std::runtime_error("reserved offering without a reservation id");
          if (k_rid < 0) k_rid = D.key(kReservationID);
          ridx = D.value(k_rid, rid);
          const int cap = (int)of.at("reservationCapacity").i(0);
          if (ridx >= (int)resv_capacity.size()) resv_capacity.resize(ridx + 1, INT32_MAX);
          if (cap < resv_capacity[ridx]) resv_capacity[ridx] = cap;
        } else if (!rid.empty()) throw Unsupported("reservation id on a non-reserved offering");
        it_offs[i].push_back({D.value(k_zone, zone), D.value(k_ct, ct), of.at("price").d(), of.at("available").boolean_or(true), ridx, cap_ov, has_oh, oh_ov});
      }
    const int n_resv = (int)resv_capacity.size();
    if (n_resv > 64) throw Unsupported("more than 64 capacity reservations");
    const int n_zones = (int)D.values[k_zone].size(), n_cts = (int)D.values[k_ct].size();
    if (n_zones > KSOLVE_MAX_ZONES || n_cts > KSOLVE_MAX_CAPTYPES) throw Unsupported("more than 16 offering zones or 4 capacity types");
    for (int i = 0; i < n_its; ++i) { it_exprs[i] = parse_exprs(its_json[i].at("requirements")); for (auto& e : it_exprs[i]) D.note(e); }

    trace("node pools");
    // ---- node pools -> templates (OrderByWeight; static pools and pools without instance types are skipped) ----
    struct Pool { const Value* v; std::string name; int weight; };
    std::vector<Pool> pools;
    for (auto& np : root.at("nodePools").items()) {
      if (np.at("static").boolean_or(false)) continue;
      if (np.has("instanceTypes") && !np.at("instanceTypes").is_null() && np.at("instanceTypes").items().empty()) continue;
      pools.push_back({&np, np.at("name").s(), (int)np.at("weight").i(0)});
    }
    std::sort(pools.begin(), pools.end(), [](const Pool& a, const Pool& b) { return a.weight != b.weight ? a.weight > b.weight : a.name > b.name; });
    const int n_templates = (int)pools.size();
    if (n_templates > KSOLVE_MAX_TEMPLATES) throw Unsupported("more than 32 NodePools");
    std::vector<std::vector<Expr>> tmpl_exprs(n_templates);
    std::vector<Taint> distinct_taints;
    auto taint_id = [&](const Taint& t) {
      for (size_t i = 0; i < distinct_taints.size(); ++i) if (!(distinct_taints[i] < t) && !(t < distinct_taints[i])) return (int)i;
      distinct_taints.push_back(t);
      if (distinct_taints.size() > 64) throw Unsupported("more than 64 distinct taints");
      return (int)distinct_taints.size() - 1;
    };
    std::vector<uint64_t> tmpl_taints(n_templates, 0);
    bool tolerate_prefer_no_schedule = false;
    struct AnyPool { const Value* v; uint64_t taints; };
    std::vector<AnyPool> all_pools;   // every non-static NodePool, also those without instance types (topology.go:105-146, scheduler.go:139-147)
    for (auto& np : root.at("nodePools").items()) {
      if (np.at("static").boolean_or(false)) continue;
      uint64_t m = 0;
      for (auto& tv : np.at("taints").items()) {
        Taint x{tv.at("key").s(), tv.at("value").s(), tv.at("effect").s()};
        m |= 1ull << taint_id(x);
        if (x.effect == "PreferNoSchedule") tolerate_prefer_no_schedule = true;
      }
      for (auto& e : parse_exprs(np.at("requirements"))) D.note(e);
      for (auto& e : label_exprs(np.at("labels"))) D.note(e);
      all_pools.push_back({&np, m});
    }
    for (int t = 0; t < n_templates; ++t) {
      const Value& np = *pools[t].v;
      tmpl_exprs[t] = parse_exprs(np.at("requirements"));
      for (auto& e : label_exprs(np.at("labels"))) tmpl_exprs[t].push_back(e);
      tmpl_exprs[t].push_back(Expr{kNodePool, "In", {pools[t].name}, -1});
      tmpl_exprs[t].push_back(Expr{np.at("nodeClassLabelKey").s("karpenter.test.sh/testnodeclass"), "In", {np.at("nodeClassName").s("default")}, -1});
      tmpl_exprs[t].push_back(Expr{"karpenter.sh/registered", "In", {"true"}, -1});
      tmpl_exprs[t].push_back(Expr{"karpenter.sh/initialized", "In", {"true"}, -1});
      for (auto& e : tmpl_exprs[t]) D.note(e);
      for (auto& tv : np.at("taints").items()) {
        Taint x{tv.at("key").s(), tv.at("value").s(), tv.at("effect").s()};
        tmpl_taints[t] |= 1ull << taint_id(x);
        if (x.effect == "PreferNoSchedule") tolerate_prefer_no_schedule = true;
      }
    }

    trace("pods");
    // ---- pods (explicit list + deterministic groups) ----
    std::vector<std::pair<std::string, std::map<std::string, std::string>>> namespace_lister;
    for (auto& nv : root.at("namespaces").items()) {
      std::map<std::string, std::string> labels;
      for (auto& kv : nv.at("labels").members()) labels[kv.first] = kv.second.s();
      namespace_lister.push_back({nv.at("name").s(), labels});
    }
    t_namespace_lister = &namespace_lister;   // read by parse_aff_term for every pod parsed below (this thread only)
    struct ListerScope { ~ListerScope() { t_namespace_lister = nullptr; } } lister_scope;   // never leave it pointing at this frame
    struct Row { int spec; };  // index into specs
    std::vector<PodSpec> specs;           // distinct pod templates (each explicit pod is its own spec)
    std::vector<int> pod_spec;            // per pod -> spec
    std::vector<uint64_t> uid_hi, uid_lo;
    UidTexts uid_text;    // explicit pods only (group pods: regenerated on demand)
    std::vector<std::pair<uint64_t, uint64_t>> group_of_pod;  // (seed, index) for group pods
    bool all_uuid = true;
    std::vector<int> pod_node_input;      // group pods bound to a node: index into the problem's stateNodes list, else -1
    for (auto& pv : root.at("pods").items()) {
      specs.push_back(parse_pod(pv));
      pod_spec.push_back((int)specs.size() - 1);
      uint64_t hi = 0, lo = 0;
      if (!parse_uuid(specs.back().uid, hi, lo)) all_uuid = false;
      uid_hi.push_back(hi); uid_lo.push_back(lo); uid_text.push_back(specs.back().uid); group_of_pod.push_back({0, 0});
      pod_node_input.push_back(-1);
    }
    struct GroupSpan { size_t at; long long cnt; uint64_t seed; std::vector<int32_t> node; };
    std::vector<GroupSpan> spans;
    for (auto& g : root.at("podGroups").items()) {
      specs.push_back(parse_pod(g.at("template")));
      int si = (int)specs.size() - 1;
      uint64_t seed = (uint64_t)g.at("uidSeed").i(0);
      long long cnt = g.at("count").i(0);
      if (cnt < 0) throw std::runtime_error("podGroups[].count is negative");
      // bound pods of a resident cluster: pod i of the group runs on stateNodes[nodeIndex[i]] (a 2M-pod cluster is a few
      // hundred templates and one integer per pod); nodeIndexB64 = the same list as base64 of little-endian int32
      std::vector<int32_t> gnode;
      if (g.has("nodeIndexB64") && !g.at("nodeIndexB64").is_null()) gnode = b64_int32(g.at("nodeIndexB64").s());
      else if (g.has("nodeIndex") && !g.at("nodeIndex").is_null()) for (auto& x : g.at("nodeIndex").items()) gnode.push_back((int32_t)x.i(-1));
      const bool bound = !gnode.empty();
      if (bound && (long long)gnode.size() != cnt) throw std::runtime_error("podGroups[].nodeIndex must have one entry per pod");
      // one Solve() of the reference handles a batch of pending pods; 16M pods is 16x the largest BASELINE configuration
      if (cnt > kMaxPodsPerProblem || (long long)pod_spec.size() + cnt > kMaxPodsPerProblem) throw Unsupported("more than 16777216 pods in one problem");
      const size_t at = pod_spec.size();
      pod_spec.resize(at + (size_t)cnt, si);
      spans.push_back(GroupSpan{at, cnt, seed, std::move(gnode)});
    }
    // every group pod's uid and binding: one pass over all groups on a few threads (a 1M-pod batch is ~600 groups)
    uid_hi.resize(pod_spec.size()); uid_lo.resize(pod_spec.size()); group_of_pod.resize(pod_spec.size()); pod_node_input.resize(pod_spec.size(), -1);
    {
      std::vector<size_t> starts;
      for (auto& sp : spans) starts.push_back(sp.at);
      const size_t first = spans.empty() ? pod_spec.size() : spans[0].at;
      parallel_for(pod_spec.size() - first, [&](size_t k) {
        const size_t p = first + k;
        const size_t gi = (size_t)(std::upper_bound(starts.begin(), starts.end(), p) - starts.begin()) - 1;
        const GroupSpan& sp = spans[gi];
        const uint64_t i = (uint64_t)(p - sp.at);
        uint64_t hi, lo;
        group_uid(sp.seed, i, hi, lo, nullptr);
        uid_hi[p] = hi; uid_lo[p] = lo; group_of_pod[p] = {sp.seed, i};
        if (!sp.node.empty()) pod_node_input[p] = (int)sp.node[(size_t)i];
      });
    }
    const int n_pods = (int)pod_spec.size();
    if (!all_uuid) {
      // UIDs that are not lower-case UUIDs: order them as strings on the host and hand the device their rank
      std::vector<int> order(n_pods);
      for (int i = 0; i < n_pods; ++i) order[i] = i;
      auto text = [&](int i) { if (!uid_text[i].empty()) return uid_text[i]; std::string s; uint64_t a, b; group_uid(group_of_pod[i].first, group_of_pod[i].second, a, b, &s); return s; };
      std::vector<std::string> texts(n_pods);
      for (int i = 0; i < n_pods; ++i) texts[i] = text(i);
      std::sort(order.begin(), order.end(), [&](int a, int b) { return texts[a] < texts[b]; });
      for (int r = 0; r < n_pods; ++r) { uid_hi[order[r]] = 0; uid_lo[order[r]] = (uint64_t)r; }
    }
    // requirement ladders per spec: row 0 = as submitted, then one row per Preferences.Relax step (preferences.go:38-57)
    struct Variant { std::vector<Expr> reqs, strict; std::vector<Toleration> tolerations; PodSpec pod; };
    std::vector<std::vector<Variant>> ladders(specs.size());
    for (size_t si = 0; si < specs.size(); ++si) {
      PodSpec p = specs[si];
      for (;;) {
        // newPodRequirements — requirements.go:91-118
        auto build = [&](bool required_only) {
          std::vector<Expr> out = label_exprs(p.node_selector);
          if (!p.has_node_affinity) return out;
          if (!required_only && !p.preferred.empty()) {
            small_stable_sort(p.preferred, [](const std::pair<int, std::vector<Expr>>& a, const std::pair<int, std::vector<Expr>>& b) { return a.first > b.first; });
            for (auto& e : p.preferred[0].second) out.push_back(e);
          }
          if (p.has_required && !p.required_terms.empty()) for (auto& e : p.required_terms[0]) out.push_back(e);
          return out;
        };
        Variant v;
        v.reqs = build(ignore_prefs);
        v.strict = (p.has_node_affinity && !p.preferred.empty()) ? build(true) : v.reqs;   // scheduler.go:561-566
        v.tolerations = p.tolerations;
        v.pod = p;
        for (auto& e : v.reqs) D.note(e);
        for (auto& e : v.strict) D.note(e);
        for (auto& term : p.required_terms) for (auto& e : term) D.note(e);   // every term feeds the TopologyNodeFilter (topologynodefilter.go:50-62)
        for (auto& alt : p.volume_requirements) for (auto& e : alt) D.note(e);
        for (auto& t : p.tscs) if (t.key != kHostname) D.key(t.key);
        for (auto* list : {&p.aff_required, &p.anti_required}) for (auto& t : *list) if (t.key != kHostname) D.key(t.key);
        for (auto* list : {&p.aff_preferred, &p.anti_preferred}) for (auto& t : *list) if (t.second.key != kHostname) D.key(t.second.key);
        ladders[si].push_back(v);
        // Relax: first relaxation that applies (preferences.go:38-57)
        if (p.has_node_affinity && p.has_required && p.required_terms.size() > 1) { p.required_terms.erase(p.required_terms.begin()); continue; }
        auto by_weight = [](const std::pair<int, AffTerm>& a, const std::pair<int, AffTerm>& b) { return a.first > b.first; };
        if (p.has_pod_affinity && !p.aff_preferred.empty()) {     // removePreferredPodAffinityTerm — preferences.go:89-101
          std::stable_sort(p.aff_preferred.begin(), p.aff_preferred.end(), by_weight);
          p.aff_preferred.erase(p.aff_preferred.begin());
          continue;
        }
        if (p.has_pod_anti && !p.anti_preferred.empty()) {        // removePreferredPodAntiAffinityTerm — preferences.go:103-115
          std::stable_sort(p.anti_preferred.begin(), p.anti_preferred.end(), by_weight);
          p.anti_preferred.erase(p.anti_preferred.begin());
          continue;
        }
        if (p.has_node_affinity && !p.preferred.empty()) {
          std::stable_sort(p.preferred.begin(), p.preferred.end(), [](const std::pair<int, std::vector<Expr>>& a, const std::pair<int, std::vector<Expr>>& b) { return a.first > b.first; });
          p.preferred.erase(p.preferred.begin());
          continue;
        }
        {
          // removeTopologySpreadScheduleAnyway — preferences.go:59-73 (swap with the last element, then truncate)
          bool removed = false;
          for (size_t i = 0; i < p.tscs.size(); ++i)
            if (p.tscs[i].when == "ScheduleAnyway") { p.tscs[i] = p.tscs.back(); p.tscs.pop_back(); removed = true; break; }
          if (removed) continue;
        }
        if (tolerate_prefer_no_schedule) {
          bool have = false;
          for (auto& t : p.tolerations) if (t.key.empty() && t.op == "Exists" && t.value.empty() && t.effect == "PreferNoSchedule") have = true;
          if (!have) { p.tolerations.push_back({"", "Exists", "", "PreferNoSchedule"}); continue; }
        }
        break;
      }
    }
    trace("daemonset pods");
    // ---- daemonset pods: only their scheduling constraints and requests matter (scheduler.go:972-1043) ----
    std::vector<PodSpec> daemons;
    for (auto& dv : root.at("daemonSetPods").items()) {
      daemons.push_back(parse_pod(dv));
      const PodSpec& dp = daemons.back();
      for (auto& e : label_exprs(dp.node_selector)) D.note(e);
      for (auto& term : dp.required_terms) for (auto& e : term) D.note(e);
    }
    if (daemons.size() > 64) throw Unsupported("more than 64 daemonset pods");
    trace("existing nodes");
    // ---- existing nodes (state.StateNode read accessors): sortExistingNodes order (scheduler.go:845-858) ----
    struct NodeIn { const Value* v; std::string name, hostname; bool initialized; };
    std::vector<NodeIn> nodes;
    for (auto& nv : root.at("stateNodes").items()) {
      NodeIn n{&nv, nv.at("name").s(), "", nv.at("initialized").boolean_or(true)};
      n.hostname = nv.has("hostname") ? nv.at("hostname").s() : (nv.at("labels").has(kHostname) ? nv.at("labels").at(kHostname).s() : n.name);
      nodes.push_back(n);
    }
    std::stable_sort(nodes.begin(), nodes.end(), [](const NodeIn& a, const NodeIn& b) { if (a.initialized != b.initialized) return a.initialized; return a.name < b.name; });
    const int n_nodes = (int)nodes.size();
    // the existing node every pod is bound to (sorted index), -1 = none: explicit pods by nodeName, group pods by nodeIndex
    std::vector<int32_t> pod_node_sorted(n_pods, -1);
    std::vector<int32_t> node_input_index;
    {
      std::map<std::string, int> sorted_index;
      for (int e = 0; e < n_nodes; ++e) sorted_index[nodes[e].name] = e;
      const auto& sn = root.at("stateNodes").items();
      node_input_index.assign(sn.size(), -1);
      for (size_t i = 0; i < sn.size(); ++i) { auto f = sorted_index.find(sn[i].at("name").s()); if (f != sorted_index.end()) node_input_index[i] = f->second; }
      std::vector<int> spec_node(specs.size(), -2);
      for (int p = 0; p < n_pods; ++p) {
        if (pod_node_input[p] >= 0) { pod_node_sorted[p] = pod_node_input[p] < (int)node_input_index.size() ? node_input_index[pod_node_input[p]] : -1; continue; }
        int& sn_ = spec_node[pod_spec[p]];
        if (sn_ == -2) { auto f = sorted_index.find(specs[pod_spec[p]].node_name); sn_ = f == sorted_index.end() ? -1 : f->second; }
        pod_node_sorted[p] = sn_;
      }
    }
    // options.residentCluster: the problem is a whole cluster — bound pods are pod rows, every simulation is a probe (ksolve_sweep)
    const bool resident = opts.at("residentCluster").boolean_or(false);
    std::vector<std::vector<Expr>> node_exprs(n_nodes);
    std::vector<uint64_t> node_taints(n_nodes, 0);
    // Every requirement source other than the nodes has been noted by now. When none of them mentions kubernetes.io/hostname
    // — no pod, NodePool, instance type or daemonset selects on it — a node's own hostname requirement (existingnode.go:72)
    // can never meet another requirement on that key (hostname topology groups count per node index, not per dictionary
    // value), so it is left out instead of spending one dictionary value per node (a 10k-node cluster would need 157 words).
    const bool hostname_selected = D.key_index.count(kHostname) != 0;
    for (int e = 0; e < n_nodes; ++e) {
      node_exprs[e] = label_exprs(nodes[e].v->at("labels"));
      // NewExistingNode adds hostname In [HostName()] (existingnode.go:72); drop a hostname label so it is not intersected twice
      node_exprs[e].erase(std::remove_if(node_exprs[e].begin(), node_exprs[e].end(), [](const Expr& x) { return x.key == kHostname; }), node_exprs[e].end());
      if (hostname_selected && nodes[e].v->at("labels").has(kHostname)) node_exprs[e].push_back(Expr{kHostname, "In", {nodes[e].v->at("labels").at(kHostname).s()}, -1});
      if (hostname_selected) node_exprs[e].push_back(Expr{kHostname, "In", {nodes[e].hostname}, -1});
      for (auto& x : node_exprs[e]) D.note(x);
      for (auto& tv : nodes[e].v->at("taints").items()) node_taints[e] |= 1ull << taint_id(Taint{tv.at("key").s(), tv.at("value").s(), tv.at("effect").s()});
    }

    trace("resources");
    // ---- resources: dimensions and exact scales ----
    std::vector<std::string> res_names = {"cpu", "memory"};
    auto add_res = [&](const std::string& r) { if (r == "nodes") return; if (std::find(res_names.begin(), res_names.end(), r) == res_names.end()) res_names.push_back(r); };
    add_res("pods");
    std::vector<std::map<std::string, i128>> it_cap(n_its), it_over(n_its);
    for (int i = 0; i < n_its; ++i) {
      it_cap[i] = parse_resources(its_json[i].at("capacity")); it_over[i] = parse_resources(its_json[i].at("overhead"));
      for (auto& kv : it_cap[i]) add_res(kv.first);
      for (auto& o : it_offs[i]) for (auto& kv : o.cap_ov) add_res(kv.first);   // lo.Assign(capacity, override) may add keys (types.go:274-277)
    }
    for (auto& s : specs) for (auto& kv : s.requests) add_res(kv.first);
    for (auto& dp : daemons) for (auto& kv : dp.requests) add_res(kv.first);
    std::vector<std::map<std::string, i128>> tmpl_limits(n_templates);
    std::vector<bool> tmpl_has_limits(n_templates, false);
    for (int t = 0; t < n_templates; ++t) {
      const Value& np = *pools[t].v;
      if (np.has("limits") && !np.at("limits").is_null()) { tmpl_has_limits[t] = true; tmpl_limits[t] = parse_resources(np.at("limits")); for (auto& kv : tmpl_limits[t]) add_res(kv.first); }
    }
    std::vector<std::map<std::string, i128>> node_avail(n_nodes), node_cap(n_nodes), node_ds(n_nodes);
    for (int e = 0; e < n_nodes; ++e) {
      node_avail[e] = parse_resources(nodes[e].v->at("available")); node_cap[e] = parse_resources(nodes[e].v->at("capacity"));
      node_ds[e] = parse_resources(nodes[e].v->at("daemonSetRequests"));
      for (auto& kv : node_avail[e]) add_res(kv.first);
    }
    const int n_res = (int)res_names.size();
    if (n_res > KSOLVE_MAX_RES) throw Unsupported("more than 8 resource dimensions");
    // nano-units per device unit: the greatest common divisor of every quantity of the dimension, so that the device
    // integers are exact and as small as possible (with Mi-granular memory and milli-cpu they fit 32 bits, which lets the
    // lite engine keep its instance-type tables as int32 registers)
    std::vector<i128> scale(n_res, 0);
    auto gcd128 = [](i128 a, i128 b) { while (b) { i128 t = a % b; a = b; b = t; } return a; };
    auto consider = [&](const std::map<std::string, i128>& m) {
      for (auto& kv : m) {
        if (kv.first == "nodes") continue;
        int r = (int)(std::find(res_names.begin(), res_names.end(), kv.first) - res_names.begin());
        if (r >= n_res) continue;   // a key only the overhead names: resources.Subtract keeps capacity's keys (resources.go:83-97)
        i128 v = kv.second < 0 ? -kv.second : kv.second;
        if (v) scale[r] = gcd128(scale[r], v);
      }
    };
    consider({{"pods", (i128)1000000000}});   // every pod requests pods: 1 (resources.go:30-38)
    for (int i = 0; i < n_its; ++i) { consider(it_cap[i]); consider(it_over[i]); }
    for (int i = 0; i < n_its; ++i) for (auto& kv : it_cap[i]) if (kv.first.rfind("hugepages-", 0) == 0) consider({{"memory", kv.second}});  // subtracted from memory
    for (int i = 0; i < n_its; ++i) for (auto& o : it_offs[i]) {
      consider(o.cap_ov); consider(o.oh_ov);
      for (auto& kv : o.cap_ov) if (kv.first.rfind("hugepages-", 0) == 0) consider({{"memory", kv.second}});
    }
    for (auto& s : specs) consider(s.requests);
    for (auto& dp : daemons) consider(dp.requests);
    for (int e = 0; e < n_nodes; ++e) consider(node_ds[e]);
    for (int t = 0; t < n_templates; ++t) consider(tmpl_limits[t]);
    for (int e = 0; e < n_nodes; ++e) { consider(node_avail[e]); consider(node_cap[e]); }
    for (int r = 0; r < n_res; ++r) if (scale[r] == 0) scale[r] = 1000000000;
    auto to_dev = [&](int r, i128 nano) {
      if (nano % scale[r] != 0) throw std::runtime_error("internal: a quantity of " + res_names[r] + " was not part of the scale computation");
      i128 v = nano / scale[r];
      if (v > (i128)(INT64_MAX / 4) || v < -(i128)(INT64_MAX / 4)) throw Unsupported("resource quantity does not fit the device's exact int64 encoding");
      return (int64_t)v;
    };
    auto res_get = [&](const std::map<std::string, i128>& m, const std::string& k) { auto it = m.find(k); return it == m.end() ? (i128)0 : it->second; };

    trace("dictionary complete");
    // ---- dictionary is complete ----
    const int it_words = std::max(1, (n_its + 63) / 64);
    fl.finalize_dictionary(it_words);
    const int nk = fl.kd.n_keys, rw = fl.kd.req_words;

    // newPodRequirements(pod, required only) of a daemonset pod — requirements.go:74-118
    auto daemon_reqs = [&](const PodSpec& dp) {
      ks::ReqBuf b;
      Flattener::clear(b);
      std::vector<Expr> ex = label_exprs(dp.node_selector);
      if (dp.has_node_affinity && dp.has_required && !dp.required_terms.empty()) for (auto& e : dp.required_terms[0]) ex.push_back(e);
      for (auto& e : ex) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
      return b;
    };
    // instance type tables
    std::vector<int64_t> it_alloc((size_t)n_res * n_its), it_capv((size_t)n_res * n_its);
    std::vector<uint64_t> it_avail(n_its, 0);
    std::vector<double> it_price((size_t)n_its * 64, 0.0);
    ReqTableBuilder it_reqs;
    it_reqs.init(n_its, rw, nk);
    // computeAllocatable (types.go:271-294): capacity - overhead over capacity's keys (resources.Subtract, resources.go:83-97),
    // then hugepage reservations come out of the allocatable memory, one clamp at zero per hugepage size (capacity is a
    // map, visited here in name order — the clamps commute)
    auto compute_alloc = [&](const std::map<std::string, i128>& cap, const std::map<std::string, i128>& over, int64_t* out, size_t stride) {
      for (int r = 0; r < n_res; ++r) {
        i128 alloc = cap.count(res_names[r]) ? res_get(cap, res_names[r]) - res_get(over, res_names[r]) : 0;
        if (r == 1) for (auto& kv : cap) if (kv.first.rfind("hugepages-", 0) == 0) { alloc -= kv.second; if (alloc < 0) alloc = 0; }
        out[(size_t)r * stride] = to_dev(r, alloc);
      }
    };
    // offering override groups (groupOfferingsByOverride, types.go:224-269): available offerings with a non-empty
    // CapacityOverride or an OverheadOverride, grouped by the override pair in first-seen order behind the base group
    std::vector<uint64_t> it_base_avail(n_its, 0), xg_avail;
    std::vector<uint32_t> xg_it;
    std::vector<std::vector<int64_t>> xg_alloc_rows;   // per group: n_res values
    for (int i = 0; i < n_its; ++i) {
      for (int r = 0; r < n_res; ++r) it_capv[(size_t)r * n_its + i] = to_dev(r, res_get(it_cap[i], res_names[r]));
      compute_alloc(it_cap[i], it_over[i], &it_alloc[i], n_its);
      std::vector<const Off*> first;   // the offering that defines extra group g of this type
      const size_t g0 = xg_it.size();
      for (auto& o : it_offs[i]) {
        int cell = o.zone * 4 + o.ct;
        if (o.available) {
          if ((it_avail[i] >> cell) & 1) { if (o.price < it_price[(size_t)i * 64 + cell]) it_price[(size_t)i * 64 + cell] = o.price; }
          else { it_avail[i] |= 1ull << cell; it_price[(size_t)i * 64 + cell] = o.price; }
          if (o.cap_ov.empty() && !o.has_oh) { it_base_avail[i] |= 1ull << cell; continue; }
          size_t g = 0;
          for (; g < first.size(); ++g) if (first[g]->cap_ov == o.cap_ov && first[g]->has_oh == o.has_oh && first[g]->oh_ov == o.oh_ov) break;
          if (g == first.size()) {
            first.push_back(&o);
            std::map<std::string, i128> cap = it_cap[i], over = it_over[i];
            for (auto& kv : o.cap_ov) cap[kv.first] = kv.second;          // lo.Assign replaces whole keys
            if (o.has_oh) for (auto& kv : o.oh_ov) over[kv.first] = kv.second;
            std::vector<int64_t> row(n_res);
            compute_alloc(cap, over, row.data(), 1);
            xg_it.push_back((uint32_t)i); xg_avail.push_back(0); xg_alloc_rows.push_back(row);
          }
          xg_avail[g0 + g] |= 1ull << cell;
        }
      }
      ks::ReqBuf b;
      Flattener::clear(b);
      for (auto& e : it_exprs[i]) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
      it_reqs.put(i, b);
    }
    std::vector<uint32_t> it_resv_first(n_its + 1, 0);
    std::vector<uint8_t> resv_zone, resv_id;
    std::vector<double> resv_price;
    for (int i = 0; i < n_its; ++i) {
      for (auto& o : it_offs[i]) if (o.rid >= 0 && o.available) { resv_zone.push_back((uint8_t)o.zone); resv_id.push_back((uint8_t)o.rid); resv_price.push_back(o.price); }
      it_resv_first[i + 1] = (uint32_t)resv_zone.size();
    }
    if (resv_zone.empty()) { resv_zone.push_back(0); resv_id.push_back(0); resv_price.push_back(0); }
    // templates
    ReqTableBuilder tmpl_reqs;
    tmpl_reqs.init(n_templates, rw, nk);
    std::vector<uint64_t> tmpl_its((size_t)n_templates * it_words, 0);
    std::vector<uint32_t> tmpl_limit_mask(n_templates, 0);
    std::vector<int64_t> tmpl_lim((size_t)n_templates * (n_res + 1), 0);
    S->node_tmpl.assign(n_nodes, -1);
    S->node_limit_cap.assign((size_t)n_nodes * (n_res + 1), 0);
    for (int t = 0; t < n_templates; ++t) {
      ks::ReqBuf b;
      Flattener::clear(b);
      for (auto& e : tmpl_exprs[t]) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
      tmpl_reqs.put(t, b);
      const Value& np = *pools[t].v;
      if (np.has("instanceTypes") && !np.at("instanceTypes").is_null()) {
        int prev = -1;
        for (auto& n : np.at("instanceTypes").items()) {
          auto f = it_index.find(n.s());
          if (f == it_index.end()) throw std::runtime_error("unknown instance type " + n.s());
          if (f->second < prev) throw Unsupported("NodePool instance types must be listed in catalogue order");
          prev = f->second;
          tmpl_its[(size_t)t * it_words + f->second / 64] |= 1ull << (f->second % 64);
        }
      } else for (int i = 0; i < n_its; ++i) tmpl_its[(size_t)t * it_words + i / 64] |= 1ull << (i % 64);
      if (tmpl_has_limits[t]) {
        // updateRemainingResources (scheduler.go:835-842): limits minus the capacity of the pool's existing nodes
        for (int e = 0; e < n_nodes; ++e) {
          const Value& nl = nodes[e].v->at("labels");
          if (!nl.has(kNodePool) || nl.at(kNodePool).s() != pools[t].name) continue;
          for (auto& kv : tmpl_limits[t]) { auto f = node_cap[e].find(kv.first); if (f != node_cap[e].end()) kv.second -= f->second; }
        }
        for (int e = 0; e < n_nodes; ++e) {
          const Value& nl = nodes[e].v->at("labels");
          if (!nl.has(kNodePool) || nl.at(kNodePool).s() != pools[t].name) continue;
          S->node_tmpl[e] = t;
          for (auto& kv : tmpl_limits[t]) {
            auto f = node_cap[e].find(kv.first);
            if (f == node_cap[e].end()) continue;
            if (kv.first == "nodes") { S->node_limit_cap[(size_t)e * (n_res + 1) + n_res] = (int64_t)(f->second / 1000000000); continue; }
            int r = (int)(std::find(res_names.begin(), res_names.end(), kv.first) - res_names.begin());
            S->node_limit_cap[(size_t)e * (n_res + 1) + r] = to_dev(r, f->second);
          }
        }
        for (auto& kv : tmpl_limits[t]) {
          if (kv.first == "nodes") { tmpl_limit_mask[t] |= 1u << n_res; tmpl_lim[(size_t)t * (n_res + 1) + n_res] = (int64_t)(kv.second / 1000000000); continue; }
          int r = (int)(std::find(res_names.begin(), res_names.end(), kv.first) - res_names.begin());
          tmpl_limit_mask[t] |= 1u << r;
          tmpl_lim[(size_t)t * (n_res + 1) + r] = to_dev(r, kv.second);
        }
      }
    }
    // daemon-overhead groups per template (scheduler.go:972-1043): instance types keyed by the set of daemonset pods that
    // could run on a node of that type from that NodePool
    // host ports: the distinct <ip, port, protocol> triples of the problem, one bit each. They only matter when a pod to
    // be scheduled binds one; then the triples of daemon pods and of the existing nodes' bound pods join the dictionary.
    std::vector<HostPort> hp_dict;
    bool hp_on = false;
    for (auto& sp : specs) hp_on = hp_on || !sp.host_ports.empty();
    auto hp_mask = [&](const std::vector<HostPort>& ports) {
      uint64_t m = 0;
      for (auto& h : ports) {
        size_t i = 0;
        for (; i < hp_dict.size(); ++i) if (hp_dict[i] == h) break;
        if (i == hp_dict.size()) { if (hp_dict.size() == 64) throw Unsupported("more than 64 distinct host ports"); hp_dict.push_back(h); }
        m |= 1ull << i;
      }
      return m;
    };
    std::vector<uint64_t> spec_hp(specs.size(), 0), daemon_hp(daemons.size(), 0), node_hp(n_nodes, 0), dg_hp;
    if (hp_on) {
      for (size_t si = 0; si < specs.size(); ++si) spec_hp[si] = hp_mask(specs[si].host_ports);
      for (size_t di = 0; di < daemons.size(); ++di) daemon_hp[di] = hp_mask(daemons[di].host_ports);
      for (int e = 0; e < n_nodes; ++e) node_hp[e] = hp_mask(parse_host_ports(nodes[e].v->at("hostPorts")));
    }
    auto hp_conflicts = [&](uint64_t use) {   // every triple of the dictionary that Matches one of `use`
      uint64_t m = 0;
      for (size_t i = 0; i < hp_dict.size(); ++i) if ((use >> i) & 1) for (size_t j = 0; j < hp_dict.size(); ++j) if (hp_dict[i].matches(hp_dict[j])) m |= 1ull << j;
      return m;
    };
    std::vector<uint32_t> dg_first;
    std::vector<uint64_t> dg_its;
    std::vector<int64_t> dg_ov;
    std::vector<uint8_t> dg_nonempty;
    if (!daemons.empty()) {
      struct DTerm { std::vector<ks::ReqBuf> reqs; };
      std::vector<DTerm> dterms(daemons.size());
      for (size_t di = 0; di < daemons.size(); ++di) {
        PodSpec p = daemons[di];
        for (;;) {   // isDaemonPodCompatible relaxes required node-affinity terms one by one (scheduler.go:1029-1043)
          dterms[di].reqs.push_back(daemon_reqs(p));
          if (p.has_node_affinity && p.has_required && p.required_terms.size() > 1) { p.required_terms.erase(p.required_terms.begin()); continue; }
          break;
        }
      }
      dg_first.push_back(0);
      for (int t = 0; t < n_templates; ++t) {
        ks::ReqRef tr;
        tr.mask = tmpl_reqs.mask.data() + (size_t)t * rw; tr.defined = tmpl_reqs.defined[t]; tr.complement = tmpl_reqs.complement[t];
        tr.has_gte = tmpl_reqs.has_gte[t]; tr.has_lte = tmpl_reqs.has_lte[t]; tr.gte = tmpl_reqs.gte.data() + (size_t)t * nk; tr.lte = tmpl_reqs.lte.data() + (size_t)t * nk; tr.minv = nullptr;
        std::vector<uint64_t> tolerated(daemons.size(), 0);
        for (size_t di = 0; di < daemons.size(); ++di) {
          std::vector<Toleration> tols = daemons[di].tolerations;
          tols.push_back({"", "Exists", "", "PreferNoSchedule"});
          bool ok = true;
          for (size_t ti = 0; ti < distinct_taints.size(); ++ti) if ((tmpl_taints[t] >> ti) & 1) {
            bool one = false;
            for (auto& tl : tols) one = one || tolerates(tl, distinct_taints[ti]);
            ok = ok && one;
          }
          tolerated[di] = ok;
        }
        std::vector<uint64_t> keys;   // group key = mask of compatible daemons, in order of first appearance
        const size_t base = dg_nonempty.size();
        for (int i = 0; i < n_its; ++i) {
          if (!((tmpl_its[(size_t)t * it_words + i / 64] >> (i % 64)) & 1)) continue;
          ks::ReqRef ir; ir.mask = it_reqs.mask.data() + (size_t)i * rw; ir.defined = it_reqs.defined[i]; ir.complement = it_reqs.complement[i];
          ir.has_gte = it_reqs.has_gte[i]; ir.has_lte = it_reqs.has_lte[i]; ir.gte = it_reqs.gte.data() + (size_t)i * nk; ir.lte = it_reqs.lte.data() + (size_t)i * nk; ir.minv = nullptr;
          uint64_t key = 0;
          for (size_t di = 0; di < daemons.size(); ++di) {
            if (!tolerated[di]) continue;
            for (auto& rq : dterms[di].reqs) {
              ks::ReqRef q = ks::reqbuf_ref_with_minv(rq);
              if (ks::reqs_compatible(fl.kd, tr, q, true) == ks::COMPAT_OK && ks::reqs_intersect(fl.kd, ir, q)) { key |= 1ull << di; break; }
            }
          }
          size_t gi = 0;
          for (; gi < keys.size(); ++gi) if (keys[gi] == key) break;
          if (gi == keys.size()) {
            keys.push_back(key);
            dg_its.resize(dg_its.size() + it_words, 0);
            dg_nonempty.push_back(key != 0);
            { uint64_t m = 0; for (size_t di = 0; di < daemons.size(); ++di) if ((key >> di) & 1) m |= daemon_hp[di]; dg_hp.push_back(m); }   // scheduler.go:990-993
            for (int r = 0; r < n_res; ++r) {
              i128 sum = 0;
              for (size_t di = 0; di < daemons.size(); ++di) if ((key >> di) & 1) sum += res_names[r] == "pods" ? (i128)1000000000 : res_get(daemons[di].requests, res_names[r]);
              dg_ov.push_back(to_dev(r, sum));
            }
          }
          dg_its[(base + gi) * it_words + i / 64] |= 1ull << (i % 64);
        }
        if (keys.empty()) { dg_its.resize(dg_its.size() + it_words, 0); dg_nonempty.push_back(0); dg_hp.push_back(0); for (int r = 0; r < n_res; ++r) dg_ov.push_back(0); }
        dg_first.push_back((uint32_t)dg_nonempty.size());
      }
      if (dg_nonempty.size() > 64) throw Unsupported("more than 64 daemon-overhead groups");
    }
    // existing node tables
    ReqTableBuilder node_reqs;
    node_reqs.init(n_nodes, rw, nk);
    std::vector<int64_t> node_remaining((size_t)n_res * std::max(1, n_nodes), 0);
    std::vector<uint8_t> node_init(std::max(1, n_nodes), 0), node_uca(std::max(1, n_nodes), 0);
    for (int e = 0; e < n_nodes; ++e) {
      ks::ReqBuf b;
      Flattener::clear(b);
      for (auto& x : node_exprs[e]) { ks::ReqBuf one; fl.encode(x, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
      node_reqs.put(e, b);
      // daemons that would run on the node minus what already runs there (scheduler.go:805-832, existingnode.go:50-64)
      std::map<std::string, i128> daemon;
      int n_daemons = 0;
      if (!daemons.empty()) {
        ks::ReqBuf lb;
        Flattener::clear(lb);
        for (auto& x : label_exprs(nodes[e].v->at("labels"))) {
          if (x.key == kHostname && !hostname_selected) continue;   // not in the dictionary: no daemonset selects on it
          ks::ReqBuf one; fl.encode(x, one); ks::reqbuf_add(fl.kd, lb, ks::reqbuf_ref_with_minv(one));
        }
        for (auto& dp : daemons) {
          bool tolerated = true;
          for (size_t ti = 0; ti < distinct_taints.size(); ++ti) if ((node_taints[e] >> ti) & 1) {
            bool ok = false;
            for (auto& t : dp.tolerations) ok = ok || tolerates(t, distinct_taints[ti]);
            tolerated = tolerated && ok;
          }
          if (!tolerated) continue;
          ks::ReqBuf pr = daemon_reqs(dp);
          if (ks::reqs_compatible(fl.kd, ks::reqbuf_ref_with_minv(lb), ks::reqbuf_ref_with_minv(pr), false) != ks::COMPAT_OK) continue;
          for (auto& kv : dp.requests) daemon[kv.first] += kv.second;
          n_daemons++;
        }
      }
      daemon["pods"] = (i128)n_daemons * 1000000000;
      for (auto& kv : node_ds[e]) daemon[kv.first] -= kv.second;
      for (int r = 0; r < n_res; ++r) {
        i128 dmn = res_get(daemon, res_names[r]);
        if (dmn < 0) dmn = 0;
        node_remaining[(size_t)r * n_nodes + e] = to_dev(r, res_get(node_avail[e], res_names[r]) - dmn);
      }
      node_init[e] = nodes[e].initialized ? 1 : 0;
      node_uca[e] = (opts.at("consolidationSimulation").boolean_or(false) && nodes[e].v->at("underConsolidateAfter").boolean_or(false)) ? 1 : 0;
    }
    // pod rows: rows [0,n_pods) are the pods; ladder rows are shared per spec and appended after
    std::vector<int> spec_first_extra(specs.size(), -1);
    int n_rows = n_pods;
    for (size_t si = 0; si < specs.size(); ++si) if (ladders[si].size() > 1) { spec_first_extra[si] = n_rows; n_rows += (int)ladders[si].size() - 1; }
    std::vector<int64_t> pod_requests((size_t)n_res * n_rows);
    ReqTableBuilder pod_reqs, pod_strict;
    pod_reqs.init(n_rows, rw, nk, true);
    std::vector<uint64_t> pod_tol(n_rows, 0), pod_hp(n_rows, 0), pod_hpc(n_rows, 0);
    std::vector<int32_t> pod_next(n_rows, -1);
    std::vector<int64_t> pod_creation(n_pods);
    std::vector<uint8_t> pod_pending(n_pods);
    // encode each (spec, variant) once, then replicate to its rows
    struct Enc { ks::ReqBuf reqs, strict; uint64_t tol; std::vector<int64_t> req; };
    std::vector<std::vector<Enc>> enc(specs.size());
    for (size_t si = 0; si < specs.size(); ++si) {
      std::vector<int64_t> req(n_res);
      for (int r = 0; r < n_res; ++r) req[r] = to_dev(r, res_names[r] == "pods" ? (i128)1000000000 : res_get(specs[si].requests, res_names[r]));
      for (auto& v : ladders[si]) {
        Enc e;
        Flattener::clear(e.reqs); Flattener::clear(e.strict);
        for (auto& x : v.reqs) { ks::ReqBuf one; fl.encode(x, one); ks::reqbuf_add(fl.kd, e.reqs, ks::reqbuf_ref_with_minv(one)); }
        for (auto& x : v.strict) { ks::ReqBuf one; fl.encode(x, one); ks::reqbuf_add(fl.kd, e.strict, ks::reqbuf_ref_with_minv(one)); }
        e.tol = 0;
        for (size_t ti = 0; ti < distinct_taints.size(); ++ti) for (auto& t : v.tolerations) if (tolerates(t, distinct_taints[ti])) { e.tol |= 1ull << ti; break; }
        e.req = req;
        enc[si].push_back(e);
      }
    }
    // StrictRequirements (without the preferred terms, scheduler.go:217-229) differ from Requirements only for pods with
    // preferences: when no variant of any pod has one, the strict table IS the requirement table (one upload, one stream)
    bool strict_differs = false;
    for (auto& ev : enc) for (auto& e : ev) strict_differs = strict_differs || memcmp(&e.reqs, &e.strict, sizeof(ks::ReqBuf)) != 0;
    if (strict_differs) pod_strict.init(n_rows, rw, nk, true);
    auto put_row = [&](int row, const Enc& e) {
      for (int r = 0; r < n_res; ++r) pod_requests[(size_t)r * n_rows + row] = e.req[r];
      pod_reqs.put(row, e.reqs); if (strict_differs) pod_strict.put(row, e.strict); pod_tol[row] = e.tol;
    };
    std::vector<uint64_t> spec_hpc(specs.size(), 0);
    for (size_t si = 0; si < specs.size(); ++si) spec_hpc[si] = hp_conflicts(spec_hp[si]);
    // volume requirement alternatives (PodData.VolumeRequirements): one requirement set per alternative, equal lists shared
    bool any_volume = false;
    for (auto& sp : specs) any_volume = any_volume || !sp.volume_requirements.empty();
    std::vector<uint32_t> spec_vol_first(specs.size(), 0), spec_vol_count(specs.size(), 0), pod_vol_first, pod_vol_count;
    std::vector<ks::ReqBuf> vol_sets;
    if (any_volume) {
      std::map<std::string, std::pair<uint32_t, uint32_t>> lists;
      for (size_t si = 0; si < specs.size(); ++si) {
        auto& alts = specs[si].volume_requirements;
        if (alts.empty()) continue;
        std::string key;
        for (auto& alt : alts) {
          for (auto& e : alt) { key += e.key; key += '\x01'; key += e.op; for (auto& v : e.values) { key += '\x02'; key += v; } key += '\x03'; }
          key += '\x04';
        }
        auto it = lists.find(key);
        if (it == lists.end()) {
          const uint32_t first = (uint32_t)vol_sets.size();
          for (auto& alt : alts) {
            ks::ReqBuf b;
            Flattener::clear(b);
            for (auto& x : alt) { if (x.min_values >= 0) throw Unsupported("volume requirements with minValues"); ks::ReqBuf one; fl.encode(x, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
            vol_sets.push_back(b);
          }
          it = lists.emplace(key, std::make_pair(first, (uint32_t)alts.size())).first;
        }
        spec_vol_first[si] = it->second.first; spec_vol_count[si] = it->second.second;
      }
      pod_vol_first.assign(n_rows, 0); pod_vol_count.assign(n_rows, 0);
    }
    ReqTableBuilder vol_reqs;
    // ---- CSI volume limits of existing nodes (VolumeUsage, volumeusage.go:178-209; existingnode.go:88, :179) ----
    // Only drivers that have a limit on some node can ever reject a pod; volumes of other drivers are dropped here.
    std::vector<std::string> pv_drivers;
    std::vector<uint8_t> volume_driver;
    std::vector<uint32_t> pod_pv_first, pod_pvs, node_pv_first, node_pvs;
    std::vector<int32_t> node_pv_limit;
    {
      std::map<std::string, int> drv;
      for (int e = 0; e < n_nodes; ++e)
        for (auto& kv : nodes[e].v->at("volumeUsage").at("limits").members()) if (!drv.count(kv.first)) { const int id = (int)drv.size(); drv[kv.first] = id; pv_drivers.push_back(kv.first); }
      if (pv_drivers.size() > KSOLVE_MAX_VOLUME_DRIVERS) throw Unsupported("more than 8 CSI drivers with volume limits");
      if (!pv_drivers.empty()) {
        const int nd = (int)pv_drivers.size();
        std::map<std::pair<int, std::string>, uint32_t> vol_id;
        auto vid = [&](const std::string& d, const std::string& c) -> int64_t {
          auto f = drv.find(d);
          if (f == drv.end()) return -1;
          auto key = std::make_pair(f->second, c);
          auto g = vol_id.find(key);
          if (g != vol_id.end()) return g->second;
          const uint32_t id = (uint32_t)vol_id.size();
          vol_id[key] = id; volume_driver.push_back((uint8_t)f->second);
          return id;
        };
        std::vector<std::vector<uint32_t>> spec_pvs(specs.size());
        for (size_t si = 0; si < specs.size(); ++si) {
          for (auto& dv : specs[si].volumes) { const int64_t id = vid(dv.first, dv.second); if (id >= 0) spec_pvs[si].push_back((uint32_t)id); }
          std::sort(spec_pvs[si].begin(), spec_pvs[si].end());
          spec_pvs[si].erase(std::unique(spec_pvs[si].begin(), spec_pvs[si].end()), spec_pvs[si].end());
        }
        pod_pv_first.assign((size_t)n_pods + 1, 0);
        for (int p = 0; p < n_pods; ++p) { pod_pv_first[p] = (uint32_t)pod_pvs.size(); pod_pvs.insert(pod_pvs.end(), spec_pvs[pod_spec[p]].begin(), spec_pvs[pod_spec[p]].end()); }
        pod_pv_first[n_pods] = (uint32_t)pod_pvs.size();
        node_pv_first.assign((size_t)n_nodes + 1, 0);
        node_pv_limit.assign((size_t)n_nodes * nd, -1);
        for (int e = 0; e < n_nodes; ++e) {
          node_pv_first[e] = (uint32_t)node_pvs.size();
          std::vector<uint32_t> ids;
          for (auto& vv : nodes[e].v->at("volumeUsage").at("volumes").items()) { const int64_t id = vid(vv.at("driver").s(), vv.at("pvc").s()); if (id >= 0) ids.push_back((uint32_t)id); }
          std::sort(ids.begin(), ids.end());
          ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
          node_pvs.insert(node_pvs.end(), ids.begin(), ids.end());
          std::vector<int> used(nd, 0);
          for (uint32_t id : ids) used[volume_driver[id]]++;
          bool over = false;
          for (auto& kv : nodes[e].v->at("volumeUsage").at("limits").members()) {
            const int dq = drv.at(kv.first);
            node_pv_limit[(size_t)e * nd + dq] = (int32_t)kv.second.i();
            if (used[dq] > kv.second.i()) over = true;
          }
          // already over a limit: ExceedsLimits fails for every pod, with or without volumes (it walks the union's drivers)
          if (over) node_remaining[(size_t)0 * n_nodes + e] = -1;
        }
        node_pv_first[n_nodes] = (uint32_t)node_pvs.size();
        if (pod_pvs.empty()) pod_pvs.push_back(0);
        if (node_pvs.empty()) node_pvs.push_back(0);
      }
    }
    vol_reqs.init((int)vol_sets.size(), rw, nk);
    for (size_t i = 0; i < vol_sets.size(); ++i) vol_reqs.put((int)i, vol_sets[i]);
    {
      bool bounds = false, minv = false;
      for (auto& ev : enc) for (auto& e : ev) { bounds = bounds || ((e.reqs.has_gte | e.reqs.has_lte | e.strict.has_gte | e.strict.has_lte) != 0); minv = minv || e.reqs.has_minv || e.strict.has_minv; }
      pod_reqs.ensure_columns(bounds, minv);
      if (strict_differs) pod_strict.ensure_columns(bounds, minv);
    }
    parallel_for((size_t)n_pods, [&](size_t pp) {     // every pod's row is its spec's encoding: independent writes
      const int p = (int)pp, si = pod_spec[p];
      put_row(p, enc[si][0]);
      pod_hp[p] = spec_hp[si]; pod_hpc[p] = spec_hpc[si];
      if (any_volume) { pod_vol_first[p] = spec_vol_first[si]; pod_vol_count[p] = spec_vol_count[si]; }
      pod_next[p] = spec_first_extra[si];
      pod_creation[p] = specs[si].creation;
      pod_pending[p] = specs[si].pending ? 1 : 0;
    });
    for (size_t si = 0; si < specs.size(); ++si) if (spec_first_extra[si] >= 0)
      for (size_t vi = 1; vi < ladders[si].size(); ++vi) {
        int row = spec_first_extra[si] + (int)vi - 1;
        put_row(row, enc[si][vi]);
        pod_hp[row] = spec_hp[si]; pod_hpc[row] = spec_hpc[si];
        if (any_volume) { pod_vol_first[row] = spec_vol_first[si]; pod_vol_count[row] = spec_vol_count[si]; }
        pod_next[row] = vi + 1 < ladders[si].size() ? row + 1 : -1;
      }

    trace("topology groups");
    // ---- topology groups (NewTopology, topology.go:68-103; Update, :162-194) --------------------------------
    // The device keeps one counter per (group, domain); which groups exist, what they select and what each pod variant
    // owns is object wrangling done here. Group identity follows TopologyGroup.Hash() (topologygroup.go:188-222):
    // key, type, namespaces, selector, maxSkew and — of the node filter — only what hashstructure can see (requirement
    // KEYS and minValues, policies, tolerations); minDomains and requirement values are not part of it.
    struct HGroup {
      int type = 0; std::string key; bool inverse = false, initial = false;
      int alias = -1;   // class of same-hash groups created by relaxation with different contents (first creator wins at solve time)
      std::set<std::string> namespaces; Selector sel; int max_skew = 0, min_domains = -1;
      std::string taint_policy, affinity_policy;
      std::vector<std::vector<Expr>> freqs; uint64_t ftol = 0;
      std::string identity;
      std::set<std::string> domains; std::map<std::string, int> counts;
      std::set<std::string> universe; std::map<std::string, int> node_regs;   // resident clusters: where a registered domain comes from
      std::string content() const {
        std::string c = identity + "|md" + std::to_string(min_domains) + "|tol" + std::to_string(ftol) + "|";
        for (auto& d : domains) c += d + ",";
        c += "|";
        for (auto& r : freqs) { for (auto& e : r) { c += e.key + " " + e.op + " ["; for (auto& v : e.values) c += v + ","; c += "];"; } c += "/"; }
        return c;
      }
    };
    std::vector<HGroup> groups;
    std::map<std::string, std::vector<int>> alias_members;
    int n_alias_classes = 0;
    std::vector<std::vector<std::vector<int>>> variant_owned(specs.size());   // group ids per (spec, variant)
    std::vector<std::vector<int>> spec_inverse_owned(specs.size());
    bool any_topology = false;
    for (auto& sp : specs) if (!sp.tscs.empty() || sp.has_pod_affinity || sp.has_pod_anti) any_topology = true;
    std::vector<PodSpec> cluster_pods;
    for (auto& cv : root.at("clusterPods").items()) cluster_pods.push_back(parse_pod(cv));
    for (auto& cp : cluster_pods) if (cp.has_pod_anti && !cp.anti_required.empty()) any_topology = true;
    auto tol_mask = [&](const std::vector<Toleration>& tols) {
      uint64_t m = 0;
      for (size_t ti = 0; ti < distinct_taints.size(); ++ti) for (auto& t : tols) if (tolerates(t, distinct_taints[ti])) { m |= 1ull << ti; break; }
      return m;
    };
    if (any_topology) {
      // buildDomainGroups — topology.go:105-146 ; TopologyDomainGroup.Insert — topologydomaingroup.go:36-57
      std::map<std::string, std::map<std::string, std::vector<uint64_t>>> dg;
      auto dg_insert = [&](const std::string& key, const std::string& dom, uint64_t taints) {
        auto& m = dg[key];
        auto it = m.find(dom);
        if (it == m.end() || taints == 0) { m[dom] = {taints}; return; }
        if (it->second[0] == 0) return;
        it->second.push_back(taints);
      };
      for (auto& ap : all_pools) {
        const Value& np = *ap.v;
        std::vector<Expr> pe = parse_exprs(np.at("requirements"));
        for (auto& e : label_exprs(np.at("labels"))) pe.push_back(e);
        ks::ReqBuf pb;
        Flattener::clear(pb);
        for (auto& e : pe) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, pb, ks::reqbuf_ref_with_minv(one)); }
        auto each_value = [&](const ks::ReqBuf& b, bool only_in, uint64_t taints) {
          for (int k = 0; k < nk; ++k) {
            if (!((b.defined >> k) & 1)) continue;
            if (only_in && (((b.complement >> k) & 1) || !ks::key_nonempty(fl.kd, b.mask, k))) continue;
            for (size_t v = 0; v < D.values[k].size(); ++v) {
              size_t pos = (size_t)fl.key_word_off[k] * 64 + v;
              if ((b.mask[pos / 64] >> (pos % 64)) & 1) dg_insert(D.keys[k], D.values[k][v], taints);   // requirement.Values(): the stored set
            }
          }
        };
        std::vector<int> members;
        if (np.has("instanceTypes") && !np.at("instanceTypes").is_null()) { for (auto& n : np.at("instanceTypes").items()) members.push_back(it_index.at(n.s())); }
        else for (int i = 0; i < n_its; ++i) members.push_back(i);
        for (int i : members) {
          ks::ReqBuf b = pb;
          ks::ReqRef ir; ir.mask = it_reqs.mask.data() + (size_t)i * rw; ir.defined = it_reqs.defined[i]; ir.complement = it_reqs.complement[i];
          ir.has_gte = it_reqs.has_gte[i]; ir.has_lte = it_reqs.has_lte[i]; ir.gte = it_reqs.gte.data() + (size_t)i * nk; ir.lte = it_reqs.lte.data() + (size_t)i * nk; ir.minv = it_reqs.minv.data() + (size_t)i * nk;
          ks::reqbuf_add(fl.kd, b, ir);
          each_value(b, false, ap.taints);
        }
        each_value(pb, true, ap.taints);
      }
      // state-node label sets (countDomains uses the raw labels, topology.go:376,441)
      std::vector<ks::ReqBuf> node_label_reqs(n_nodes);
      std::vector<std::map<std::string, std::string>> node_labels(n_nodes);
      for (int e = 0; e < n_nodes; ++e) {
        Flattener::clear(node_label_reqs[e]);
        for (auto& x : label_exprs(nodes[e].v->at("labels"))) {
          node_labels[e][x.key] = x.values[0];
          if (x.key == kHostname && !hostname_selected) continue;   // not in the dictionary: nothing selects on it (see above)
          ks::ReqBuf one; fl.encode(x, one); ks::reqbuf_add(fl.kd, node_label_reqs[e], ks::reqbuf_ref_with_minv(one));
        }
      }
      std::map<std::string, int> node_by_name;
      for (int e = 0; e < n_nodes; ++e) node_by_name[nodes[e].name] = e;
      // a resident cluster counts its bound pods although they are pod rows: a probe takes its displaced pods' share out again
      std::vector<std::vector<int32_t>> spec_bound_nodes(specs.size());
      if (resident) for (int p = 0; p < n_pods; ++p) if (pod_node_sorted[p] >= 0) spec_bound_nodes[pod_spec[p]].push_back(pod_node_sorted[p]);
      std::set<std::string> excluded;   // pods being scheduled are not counted from the cluster (topology.go:92-94)
      if (!cluster_pods.empty() && !resident) for (int p = 0; p < n_pods; ++p) {
        if (!uid_text[p].empty()) excluded.insert(uid_text[p]);
        else { std::string t; uint64_t a, b; group_uid(group_of_pod[p].first, group_of_pod[p].second, a, b, &t); excluded.insert(t); }
      }
      auto filter_matches = [&](const HGroup& g, uint64_t taints, const ks::ReqBuf& reqs) {   // topologynodefilter.go:68-96
        if (g.taint_policy == "Honor" && (taints & ~g.ftol)) return false;
        if (g.affinity_policy != "Honor" || g.freqs.empty()) return true;
        for (auto& r : g.freqs) {
          ks::ReqBuf b;
          Flattener::clear(b);
          for (auto& e : r) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
          if (ks::reqs_compatible(fl.kd, ks::reqbuf_ref_with_minv(reqs), ks::reqbuf_ref_with_minv(b), false) == ks::COMPAT_OK) return true;
        }
        return false;
      };
      // NewTopologyGroup — topologygroup.go:79-126
      auto make_group = [&](int type, const std::string& key, const PodSpec& pod, const std::set<std::string>& namespaces, const Selector& sel,
                            int max_skew, int min_domains, const std::string& taint_policy, const std::string& affinity_policy) {
        HGroup g;
        g.type = type; g.key = key; g.namespaces = namespaces; g.sel = sel; g.max_skew = max_skew; g.min_domains = min_domains;
        std::string fview;
        if (type == 0) {
          g.taint_policy = taint_policy.empty() ? "Ignore" : taint_policy;
          g.affinity_policy = affinity_policy.empty() ? "Honor" : affinity_policy;
          g.ftol = tol_mask(pod.tolerations);
          std::vector<Expr> selx = label_exprs(pod.node_selector);
          if (!pod.has_node_affinity || !pod.has_required) g.freqs.push_back(selx);
          else for (auto& term : pod.required_terms) { std::vector<Expr> r = selx; for (auto& e : term) r.push_back(e); g.freqs.push_back(r); }
          std::set<std::set<std::string>> rv;
          for (auto& r : g.freqs) { std::set<std::string> ks_; for (auto& e : r) ks_.insert(e.key); rv.insert(ks_); }
          for (auto& r : rv) { for (auto& k : r) fview += k + ","; fview += "/"; }
          std::set<std::tuple<std::string, std::string, std::string, std::string>> tv;
          for (auto& t : pod.tolerations) tv.insert({t.key, t.op, t.value, t.effect});
          fview += "|";
          for (auto& t : tv) fview += std::get<0>(t) + ":" + std::get<1>(t) + ":" + std::get<2>(t) + ":" + std::get<3>(t) + ",";
        }
        g.identity = key + "|" + std::to_string(type) + "|";
        for (auto& n : namespaces) g.identity += n + ",";
        g.identity += "|" + std::to_string(max_skew) + "|" + g.taint_policy + "|" + g.affinity_policy + "|" + fview + "|" + sel.canon();
        // ForEachDomain — topologydomaingroup.go:61-72
        const uint64_t ptol = tol_mask(pod.tolerations);
        auto dgi = dg.find(key);
        if (dgi != dg.end()) for (auto& kv : dgi->second) {
          bool ok = g.taint_policy == "Ignore";
          if (!ok) for (uint64_t taints : kv.second) if (!(taints & ~ptol)) { ok = true; break; }
          if (ok) g.domains.insert(kv.first);
        }
        g.universe = g.domains;
        return g;
      };
      // countDomains — topology.go:361-459. A resident cluster asks this of 100k nodes and 2M bound pods for every group (and once
      // per disruption pass), so the per-node facts are worked out once and the counting itself runs on integers: the node's domain
      // under a key as an id (per KEY), whether the node passes a group's node filter (per distinct FILTER), whether it has a Node.
      struct KeyDomains { std::vector<int32_t> label_val, pod_val; std::vector<std::string> names; };   // pod_val: hostname falls back to the node's name (:438-442)
      std::map<std::string, KeyDomains> key_domains;
      auto domains_of_key = [&](const std::string& key) -> KeyDomains& {
        auto f = key_domains.find(key);
        if (f != key_domains.end()) return f->second;
        KeyDomains& kd2 = key_domains[key];
        kd2.label_val.assign(n_nodes, -1); kd2.pod_val.assign(n_nodes, -1);
        std::unordered_map<std::string, int32_t> ids;
        auto id_of = [&](const std::string& v) { auto r = ids.emplace(v, (int32_t)kd2.names.size()); if (r.second) kd2.names.push_back(v); return r.first->second; };
        for (int e = 0; e < n_nodes; ++e) {
          auto it = node_labels[e].find(key);
          if (it != node_labels[e].end()) kd2.label_val[e] = kd2.pod_val[e] = id_of(it->second);
          else if (key == kHostname) kd2.pod_val[e] = id_of(nodes[e].name);
        }
        return kd2;
      };
      std::vector<char> node_has_node(n_nodes, 1);
      for (int e = 0; e < n_nodes; ++e) node_has_node[e] = nodes[e].v->at("hasNode").boolean_or(true) ? 1 : 0;
      std::map<std::string, std::vector<char>> filter_verdicts;   // by what a node filter consists of
      auto nodes_passing = [&](const HGroup& g) -> const std::vector<char>& {
        std::string sig = g.taint_policy + "|" + g.affinity_policy + "|" + std::to_string(g.ftol) + "|";
        for (auto& r : g.freqs) { for (auto& e : r) { sig += e.key + " " + e.op + " ["; for (auto& v : e.values) sig += v + ","; sig += "];"; } sig += "/"; }
        auto f = filter_verdicts.find(sig);
        if (f != filter_verdicts.end()) return f->second;
        std::vector<char>& ok = filter_verdicts[sig];
        ok.assign(n_nodes, 1);
        if (g.taint_policy == "Honor" || (g.affinity_policy == "Honor" && !g.freqs.empty())) {
          std::vector<ks::ReqBuf> enc_f;   // the filter's requirement sets, encoded once
          for (auto& r : g.freqs) { ks::ReqBuf b; Flattener::clear(b); for (auto& e : r) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); } enc_f.push_back(b); }
          for (int e = 0; e < n_nodes; ++e) {
            bool pass = !(g.taint_policy == "Honor" && (node_taints[e] & ~g.ftol));
            if (pass && g.affinity_policy == "Honor" && !enc_f.empty()) {
              pass = false;
              for (auto& b : enc_f) if (ks::reqs_compatible(fl.kd, ks::reqbuf_ref_with_minv(node_label_reqs[e]), ks::reqbuf_ref_with_minv(b), false) == ks::COMPAT_OK) { pass = true; break; }
            }
            ok[e] = pass ? 1 : 0;
          }
        }
        return ok;
      };
      auto count_domains = [&](HGroup& g) {
        KeyDomains& kd2 = domains_of_key(g.key);
        const std::vector<char>& node_ok = nodes_passing(g);
        std::vector<int32_t> regs(kd2.names.size(), 0), cnt(kd2.names.size(), 0);
        for (int e = 0; e < n_nodes; ++e) if (node_has_node[e] && node_ok[e] && kd2.label_val[e] >= 0) regs[kd2.label_val[e]]++;
        if (resident) {
          // the bound pod rows, spec by spec (a 2M-pod cluster is a few hundred specs)
          for (size_t si = 0; si < specs.size(); ++si) {
            if (spec_bound_nodes[si].empty() || !g.namespaces.count(specs[si].ns) || (!g.sel.nil && !g.sel.matches(specs[si].labels))) continue;
            if (specs[si].phase == "Failed" || specs[si].phase == "Succeeded") continue;
            for (int32_t e : spec_bound_nodes[si]) { const int32_t v = kd2.pod_val[e]; if (v >= 0 && node_ok[e]) cnt[v]++; }
          }
        }
        for (size_t v = 0; v < regs.size(); ++v) {
          if (regs[v]) { g.domains.insert(kd2.names[v]); g.node_regs[kd2.names[v]] += regs[v]; }
          if (cnt[v]) { g.counts[kd2.names[v]] += cnt[v]; g.domains.insert(kd2.names[v]); }
        }
        for (auto& cp : cluster_pods) {
          if (!g.namespaces.count(cp.ns)) continue;
          if (!g.sel.nil && !g.sel.matches(cp.labels)) continue;
          if (cp.node_name.empty() || cp.phase == "Failed" || cp.phase == "Succeeded") continue;
          if (excluded.count(cp.uid)) continue;
          auto nf = node_by_name.find(cp.node_name);
          if (nf == node_by_name.end()) continue;
          const int e = nf->second;
          std::string dom;
          auto it = node_labels[e].find(g.key);
          if (it != node_labels[e].end()) dom = it->second;
          else if (g.key == kHostname) dom = nodes[e].name;
          else continue;
          if (!filter_matches(g, node_taints[e], node_label_reqs[e])) continue;
          g.counts[dom]++; g.domains.insert(dom);
        }
      };
      auto namespace_list = [](const std::string& ns, const AffTerm& term) {
        return (term.namespaces.empty() && !term.resolved) ? std::set<std::string>{ns} : std::set<std::string>(term.namespaces.begin(), term.namespaces.end());
      };
      // inverse anti-affinity groups — topology.go:310-355
      std::vector<HGroup> inverse;
      auto inverse_for = [&](const PodSpec& pod, int node) {
        std::vector<int> owned;
        for (auto& term : pod.anti_required) {
          HGroup g = make_group(2, term.key, pod, namespace_list(pod.ns, term), term.sel, INT32_MAX, -1, "", "");
          int id = -1;
          for (size_t i = 0; i < inverse.size(); ++i) if (inverse[i].identity == g.identity) { id = (int)i; break; }
          if (id < 0) { g.inverse = true; g.initial = true; inverse.push_back(g); id = (int)inverse.size() - 1; }
          if (node >= 0) { auto it = node_labels[node].find(inverse[id].key); if (it != node_labels[node].end()) { inverse[id].counts[it->second]++; inverse[id].domains.insert(it->second); } }
          owned.push_back(id);
        }
        return owned;
      };
      for (auto& cp : cluster_pods) {
        if (!(cp.has_pod_anti && !cp.anti_required.empty())) continue;
        if (excluded.count(cp.uid)) continue;
        auto nf = node_by_name.find(cp.node_name);
        if (nf == node_by_name.end()) continue;
        inverse_for(cp, nf->second);
      }
      if (resident) for (size_t si = 0; si < specs.size(); ++si) {
        if (spec_bound_nodes[si].empty() || !(specs[si].has_pod_anti && !specs[si].anti_required.empty())) continue;
        const std::vector<int> ids = inverse_for(specs[si], -1);
        for (int id : ids) for (int32_t e : spec_bound_nodes[si]) { auto it = node_labels[e].find(inverse[id].key); if (it != node_labels[e].end()) { inverse[id].counts[it->second]++; inverse[id].domains.insert(it->second); } }
      }
      std::vector<std::vector<std::vector<int>>> variant_groups(specs.size());
      std::map<std::string, int> group_by_identity;
      for (int pass = 0; pass < 2; ++pass) {   // pass 0: NewTopology sees every pod as submitted; pass 1: groups that only relaxed variants own
        for (size_t si = 0; si < specs.size(); ++si) {
          if (pass == 0) {
            variant_groups[si].resize(ladders[si].size());
            const PodSpec& p0 = ladders[si][0].pod;
            const bool any_anti = p0.has_pod_anti && (!p0.anti_required.empty() || !p0.anti_preferred.empty());
            const bool req_anti = any_anti && !p0.anti_required.empty();
            if ((ignore_prefs && req_anti) || (!ignore_prefs && any_anti)) spec_inverse_owned[si] = inverse_for(p0, -1);
          }
          for (size_t vi = pass == 0 ? 0 : 1; vi < (pass == 0 ? 1 : ladders[si].size()); ++vi) {
            const PodSpec& p = ladders[si][vi].pod;
            std::vector<HGroup> tgs;
            for (auto& t : p.tscs) {   // newForTopologies — topology.go:461-495
              if (ignore_prefs && t.when != "DoNotSchedule") continue;
              tgs.push_back(make_group(0, t.key, p, {p.ns}, t.sel, t.max_skew, t.min_domains, t.taint_policy, t.affinity_policy));
            }
            if (p.has_pod_affinity) {  // newForAffinities — topology.go:498-538
              for (auto& t : p.aff_required) tgs.push_back(make_group(1, t.key, p, namespace_list(p.ns, t), t.sel, INT32_MAX, -1, "", ""));
              if (!ignore_prefs) for (auto& t : p.aff_preferred) tgs.push_back(make_group(1, t.second.key, p, namespace_list(p.ns, t.second), t.second.sel, INT32_MAX, -1, "", ""));
            }
            if (p.has_pod_anti) {
              for (auto& t : p.anti_required) tgs.push_back(make_group(2, t.key, p, namespace_list(p.ns, t), t.sel, INT32_MAX, -1, "", ""));
              if (!ignore_prefs) for (auto& t : p.anti_preferred) tgs.push_back(make_group(2, t.second.key, p, namespace_list(p.ns, t.second), t.second.sel, INT32_MAX, -1, "", ""));
            }
            for (auto& tg : tgs) {
              auto found = group_by_identity.find(tg.identity);
              int id = found == group_by_identity.end() ? -1 : found->second;
              if (id < 0) {
                count_domains(tg);
                tg.initial = pass == 0;
                groups.push_back(tg);
                id = (int)groups.size() - 1;
                group_by_identity[tg.identity] = id;
              } else if (!groups[id].initial) {
                // a group that first appears when some pod relaxes is created by whichever pod relaxes first, and every
                // later owner joins that one (Topology.Update looks the group up by hash, topology.go:162-194). Creators
                // that would build different contents under one hash become members of an alias class; the solver keeps
                // the member that is created first
                count_domains(tg);
                const std::string want = tg.content();
                int match = -1;
                for (int m : alias_members[tg.identity]) if (groups[m].content() == want) { match = m; break; }
                if (match < 0 && groups[id].content() == want) match = id;
                if (match < 0) {
                  auto& members = alias_members[tg.identity];
                  if (members.empty()) { members.push_back(id); groups[id].alias = n_alias_classes++; }
                  tg.initial = false; tg.alias = groups[id].alias;
                  groups.push_back(tg);
                  match = (int)groups.size() - 1;
                  members.push_back(match);
                }
                id = match;
              }
              variant_groups[si][vi].push_back(id);
            }
          }
        }
      }
      if (groups.size() + inverse.size() > KSOLVE_MAX_TOPO_GROUPS) throw Unsupported("more than 1024 topology groups");
      const size_t n_regular = groups.size();
      for (auto& g : inverse) groups.push_back(g);
      for (size_t si = 0; si < specs.size(); ++si) {
        variant_owned[si].resize(ladders[si].size());
        for (size_t vi = 0; vi < ladders[si].size(); ++vi) {
          variant_owned[si][vi] = variant_groups[si][vi];
          for (int id : spec_inverse_owned[si]) variant_owned[si][vi].push_back((int)n_regular + id);
        }
      }
    }
    const int G = (int)groups.size();
    std::vector<uint8_t> tg_type(G), tg_inverse(G), tg_initial(G), tg_fa(G), tg_ft(G);
    std::vector<int32_t> tg_key(G), tg_skew(G), tg_mind(G), tg_alias(G, -1);
    std::vector<uint32_t> tg_ffirst(G + 1, 0);
    std::vector<uint64_t> tg_ftol(G), tg_domains;
    std::vector<int32_t> tg_counts, tg_node_counts, tg_regs;
    std::vector<uint64_t> tg_universe;
    std::vector<uint16_t> value_rank((size_t)rw * 64, 0);
    std::vector<int32_t> node_host_value(std::max(1, n_nodes), -1);
    std::vector<uint64_t> pod_topo_owned, pod_topo_selected;
    ReqTableBuilder tg_freqs;
    uint32_t dom_words = 1;
    if (G) {
      for (auto& g : groups) if (g.key != kHostname) { int k = D.key_index.at(g.key); dom_words = std::max(dom_words, fl.key_word_off[k + 1] - fl.key_word_off[k]); }
      tg_domains.assign((size_t)G * dom_words, 0); tg_counts.assign((size_t)G * dom_words * 64, 0); tg_node_counts.assign((size_t)G * std::max(1, n_nodes), 0);
      if (resident) { tg_universe.assign((size_t)G * dom_words, 0); tg_regs.assign((size_t)G * dom_words * 64, 0); }
      int n_f = 0;
      for (auto& g : groups) n_f += (int)g.freqs.size();
      tg_freqs.init(std::max(1, n_f), rw, nk);
      std::unordered_map<std::string, int> node_of_hostname;   // a hostname group of a 100k-node cluster counts pods on most of them
      int fi = 0;
      for (int gi = 0; gi < G; ++gi) {
        const HGroup& g = groups[gi];
        tg_type[gi] = (uint8_t)g.type; tg_inverse[gi] = g.inverse; tg_initial[gi] = g.initial;
        tg_skew[gi] = g.max_skew; tg_mind[gi] = g.min_domains; tg_alias[gi] = g.alias;
        tg_fa[gi] = g.affinity_policy == "Honor"; tg_ft[gi] = g.taint_policy == "Honor"; tg_ftol[gi] = g.ftol;
        tg_ffirst[gi] = (uint32_t)fi;
        for (auto& r : g.freqs) {
          ks::ReqBuf b;
          Flattener::clear(b);
          for (auto& e : r) { ks::ReqBuf one; fl.encode(e, one); ks::reqbuf_add(fl.kd, b, ks::reqbuf_ref_with_minv(one)); }
          tg_freqs.put(fi++, b);
        }
        if (g.key == kHostname) {
          tg_key[gi] = -1;
          if (node_of_hostname.empty()) for (int e = n_nodes - 1; e >= 0; --e) node_of_hostname[nodes[e].hostname] = e;   // the first node of a hostname wins, as the linear search it replaces did
          for (auto& kv : g.counts) {
            auto f = node_of_hostname.find(kv.first);
            if (f == node_of_hostname.end()) throw Unsupported("pods counted on a hostname that is not a state node");
            tg_node_counts[(size_t)gi * n_nodes + f->second] += kv.second;
          }
          continue;
        }
        const int k = D.key_index.at(g.key);
        tg_key[gi] = k;
        for (auto& dom : g.domains) {
          auto vi = D.value_index[k].find(dom);
          if (vi == D.value_index[k].end()) throw std::runtime_error("topology domain " + dom + " missing from the dictionary");
          tg_domains[(size_t)gi * dom_words + vi->second / 64] |= 1ull << (vi->second % 64);
        }
        for (auto& kv : g.counts) tg_counts[(size_t)gi * dom_words * 64 + D.value_index[k].at(kv.first)] = kv.second;
        if (resident) {
          for (auto& dom : g.universe) { const int v = D.value_index[k].at(dom); tg_universe[(size_t)gi * dom_words + v / 64] |= 1ull << (v % 64); }
          for (auto& kv : g.node_regs) tg_regs[(size_t)gi * dom_words * 64 + D.value_index[k].at(kv.first)] = kv.second;
        }
      }
      tg_ffirst[G] = (uint32_t)fi;
      for (int k = 0; k < nk; ++k) {
        std::vector<int> order(D.values[k].size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return D.values[k][a] < D.values[k][b]; });
        if (order.size() > 65535) throw Unsupported("more than 65535 values under one label key");
        for (size_t r = 0; r < order.size(); ++r) value_rank[(size_t)fl.key_word_off[k] * 64 + order[r]] = (uint16_t)r;
      }
      if (fl.kd.key_hostname >= 0) for (int e = 0; e < n_nodes; ++e) {
        auto vi = D.value_index[fl.kd.key_hostname].find(nodes[e].hostname);
        if (vi != D.value_index[fl.kd.key_hostname].end()) node_host_value[e] = vi->second;
      }
      const int tw = (G + 63) / 64;
      pod_topo_owned.assign((size_t)n_rows * tw, 0); pod_topo_selected.assign((size_t)n_rows * tw, 0);
      std::vector<std::vector<uint64_t>> spec_selected(specs.size(), std::vector<uint64_t>(tw, 0));
      for (size_t si = 0; si < specs.size(); ++si)
        for (int gi = 0; gi < G; ++gi) if (groups[gi].namespaces.count(specs[si].ns) && groups[gi].sel.matches(specs[si].labels)) spec_selected[si][gi / 64] |= 1ull << (gi % 64);
      auto put_masks = [&](int row, size_t si, size_t vi) {
        for (int id : variant_owned[si][vi]) pod_topo_owned[(size_t)row * tw + id / 64] |= 1ull << (id % 64);
        for (int w = 0; w < tw; ++w) pod_topo_selected[(size_t)row * tw + w] = spec_selected[si][w];
      };
      for (int p = 0; p < n_pods; ++p) put_masks(p, (size_t)pod_spec[p], 0);
      for (size_t si = 0; si < specs.size(); ++si) if (spec_first_extra[si] >= 0)
        for (size_t vi = 1; vi < ladders[si].size(); ++vi) put_masks(spec_first_extra[si] + (int)vi - 1, si, vi);
    }

    trace("describe");
    // ---- describe & solve ----
    ksolve_problem_desc d{};
    d.abi_version = KSOLVE_ABI_VERSION;
    d.n_keys = (uint32_t)nk; d.key_word_off = fl.key_word_off.data(); d.well_known_mask = fl.kd.well_known_mask;
    d.key_instance_type = fl.kd.key_it; d.key_zone = fl.kd.key_zone; d.key_capacity_type = fl.kd.key_ct;
    d.value_int = fl.value_int.data(); d.value_is_int = fl.value_is_int.data();
    d.n_res = (uint32_t)n_res;
    d.n_its = (uint32_t)n_its; d.it_allocatable = it_alloc.data(); d.it_capacity = it_capv.data(); d.it_reqs = it_reqs.view();
    d.it_offering_avail = it_avail.data(); d.it_offering_price = it_price.data(); d.n_zones = (uint32_t)n_zones; d.n_captypes = (uint32_t)n_cts;
    std::vector<int64_t> xg_alloc((size_t)n_res * xg_it.size());
    for (size_t e = 0; e < xg_it.size(); ++e) for (int r = 0; r < n_res; ++r) xg_alloc[(size_t)r * xg_it.size() + e] = xg_alloc_rows[e][r];
    if (!xg_it.empty()) {
      if (xg_it.size() > KSOLVE_MAX_OVERRIDE_GROUPS) throw Unsupported("more than 4096 offering override groups");
      d.n_override_groups = (uint32_t)xg_it.size(); d.override_it = xg_it.data(); d.override_allocatable = xg_alloc.data();
      d.override_avail = xg_avail.data(); d.it_base_avail = it_base_avail.data();
    }
    d.n_templates = (uint32_t)n_templates; d.tmpl_reqs = tmpl_reqs.view(); d.tmpl_taints = tmpl_taints.data(); d.tmpl_its = tmpl_its.data();
    d.tmpl_limit_mask = tmpl_limit_mask.data(); d.tmpl_limits = tmpl_lim.data();
    d.n_reservations = (uint32_t)n_resv; d.reservation_capacity = resv_capacity.data(); d.key_reservation_id = k_rid;
    d.captype_reserved = D.value_index[k_ct].count("reserved") ? D.value_index[k_ct].at("reserved") : -1;
    d.it_reserved_first = it_resv_first.data(); d.reserved_zone = resv_zone.data(); d.reserved_id = resv_id.data(); d.reserved_price = resv_price.data();
    if (!dg_first.empty()) { d.tmpl_daemon_first = dg_first.data(); d.daemon_group_its = dg_its.data(); d.daemon_group_overhead = dg_ov.data(); d.daemon_group_nonempty = dg_nonempty.data(); }
    d.n_pods = (uint32_t)n_pods; d.n_pod_rows = (uint32_t)n_rows; d.pod_requests = pod_requests.data();
    d.pod_reqs = pod_reqs.view(); d.pod_strict_reqs = strict_differs ? pod_strict.view() : pod_reqs.view(); d.pod_tolerates = pod_tol.data(); d.pod_next_variant = pod_next.data();
    if (hp_on) {
      d.pod_host_ports = pod_hp.data(); d.pod_host_port_conflicts = pod_hpc.data();
      if (n_nodes) d.node_host_ports = node_hp.data();
      if (!dg_first.empty()) d.daemon_group_host_ports = dg_hp.data();
    }
    if (any_volume) {
      d.n_volume_reqs = (uint32_t)vol_sets.size(); d.volume_reqs = vol_reqs.view();
      d.pod_volume_first = pod_vol_first.data(); d.pod_volume_count = pod_vol_count.data();
    }
    d.pod_creation = pod_creation.data(); d.pod_uid_hi = uid_hi.data(); d.pod_uid_lo = uid_lo.data(); d.pod_is_pending = pod_pending.data();
    d.n_taints = (uint32_t)distinct_taints.size();
    d.key_hostname = fl.kd.key_hostname;
    d.n_nodes = (uint32_t)n_nodes; d.node_reqs = node_reqs.view(); d.node_taints = node_taints.data(); d.node_remaining = node_remaining.data();
    d.node_initialized = node_init.data(); d.node_under_consolidate_after = node_uca.data();
    std::vector<uint8_t> pod_from_deleting(std::max(1, n_pods), 0);
    {
      std::set<std::string> deleting;
      for (auto& nn : root.at("deletingNodeNames").items()) deleting.insert(nn.s());
      if (!deleting.empty()) for (int p = 0; p < n_pods; ++p) if (deleting.count(specs[pod_spec[p]].node_name)) pod_from_deleting[p] = 1;
    }
    d.pod_from_deleting_node = pod_from_deleting.data();
    if (resident) d.pod_node = pod_node_sorted.data();
    if (!pv_drivers.empty()) {
      d.n_volume_drivers = (uint32_t)pv_drivers.size(); d.n_volumes = (uint32_t)volume_driver.size(); d.volume_driver = volume_driver.data();
      d.pod_pv_first = pod_pv_first.data(); d.pod_pvs = pod_pvs.data(); d.node_pv_first = node_pv_first.data(); d.node_pvs = node_pvs.data();
      d.node_pv_limit = node_pv_limit.data();
    }
    if (G) {
      ksolve_topology& t = d.topo;
      t.n = (uint32_t)G; t.type = tg_type.data(); t.inverse = tg_inverse.data(); t.initially_active = tg_initial.data(); t.key = tg_key.data();
      t.max_skew = tg_skew.data(); t.min_domains = tg_mind.data(); t.domain_words = dom_words; t.domains = tg_domains.data();
      t.init_counts = tg_counts.data(); t.init_node_counts = tg_node_counts.data();
      t.filter_affinity_honor = tg_fa.data(); t.filter_taint_honor = tg_ft.data(); t.filter_first = tg_ffirst.data(); t.filter_reqs = tg_freqs.view();
      t.filter_tolerates = tg_ftol.data(); t.value_rank = value_rank.data(); t.node_hostname_value = node_host_value.data();
      if (n_alias_classes) { t.alias_class = tg_alias.data(); t.n_alias_classes = (uint32_t)n_alias_classes; }
      if (resident) { t.domain_universe = tg_universe.data(); t.domain_node_regs = tg_regs.data(); }
      S->n_topo_groups = G; S->n_alias_classes = n_alias_classes;
      d.pod_topo_owned = pod_topo_owned.data(); d.pod_topo_selected = pod_topo_selected.data();
    }
    ksolve_options ko{};
    ko.min_values_best_effort = opts.at("minValuesPolicy").s("Strict") == "BestEffort";
    ko.max_claims = (uint32_t)opts.at("maxClaims").i(0);
    ko.max_steps = opts.at("maxSteps").i(-1);
    ko.device = (uint32_t)opts.at("device").i(0);
    ko.lds_claim_cap = (uint32_t)opts.at("ldsClaimCap").i(0);
    ko.truncate_instance_types = (uint32_t)opts.at("truncateInstanceTypes").i(0);
    ko.reserved_capacity = opts.at("reservedCapacity").boolean_or(false) ? 1 : 0;
    ko.reserved_offering_strict = opts.at("reservedOfferingMode").s("Fallback") == "Strict" ? 1 : 0;
    { const std::string eng = opts.at("engine").s("auto"); ko.engine = eng == "general" ? 1u : eng == "cursor" ? 2u : eng == "cursor-wide" ? 3u : eng == "cursor-hbm" ? 4u : eng == "cursor-pair" ? 5u : eng == "spread" ? 6u : 0u; }

    trace("ksolve_create");
    ksolve_status st = api.create(&d, &ko, &handle);
    trace("session tables");
    if (st != KSOLVE_OK) {
      S->error = handle ? api.last_error(handle) : "ksolve_create failed";
      S->error_kind = st == KSOLVE_ERR_UNSUPPORTED ? "unsupported" : st == KSOLVE_ERR_NO_DEVICE ? "no_device" : "create";
      if (handle) { api.destroy(handle); handle = nullptr; }
      return S;
    }
    for (auto& p : pools) S->pool_names.push_back(p.name);
    for (auto& n : nodes) { S->node_names.push_back(n.name); S->node_initialized.push_back(n.initialized ? 1 : 0); }
    S->uid_text = uid_text; S->group_of_pod = group_of_pod; S->res_names = res_names; S->scale = scale;
    for (int i = 0; i < n_its; ++i) S->it_names.push_back(its_json[i].at("name").s());
    S->k_rid = k_rid;
    S->n_pods = n_pods; S->n_rows = n_rows; S->n_its = n_its; S->n_res = n_res; S->it_words = it_words;
    S->n_templates = n_templates; S->tmpl_lim = tmpl_lim; S->strict_shared = !strict_differs;
    // what ksched_sweep needs: the node of every pod, its pending / deleting flags, offerings and requirement values per type
    {
      S->node_input_index = node_input_index;
      S->pod_node = pod_node_sorted;
      S->pod_pending_flag.assign(pod_pending.begin(), pod_pending.end());
      S->pod_deleting_flag.assign(pod_from_deleting.begin(), pod_from_deleting.begin() + n_pods);
      S->it_offerings.resize(n_its);
      for (int i = 0; i < n_its; ++i) for (auto& o : it_offs[i]) S->it_offerings[i].push_back({o.zone, o.ct, o.rid, o.price, o.available});
      S->it_exprs = it_exprs;
      std::map<std::string, int> it_by_name;
      for (int i = 0; i < n_its; ++i) it_by_name.emplace(S->it_names[i], i);
      S->node_it.assign(nodes.size(), -1); S->node_price.assign(nodes.size(), 0.0); S->node_spot.assign(nodes.size(), 0);
      for (size_t e = 0; e < nodes.size(); ++e) {
        const Value& nl = nodes[e].v->at("labels");
        auto label = [&](const char* k) { return nl.has(k) ? nl.at(k).s() : std::string(); };
        const std::string ctl = label(kCapacityType), zl = label(kZone);
        S->node_spot[e] = ctl == "spot" ? 1 : 0;
        auto f = it_by_name.find(label(kInstanceType));
        if (f == it_by_name.end()) continue;
        S->node_it[e] = f->second;
        auto zi = D.value_index[k_zone].find(zl); auto ci = D.value_index[k_ct].find(ctl);
        if (zi == D.value_index[k_zone].end() || ci == D.value_index[k_ct].end()) continue;
        for (auto& o : it_offs[f->second]) if (o.zone == zi->second && o.ct == ci->second) { S->node_price[e] = o.price != o.price ? 0.0 : o.price; break; }
      }
    }
    trace("done");
    return S;
  } catch (const Unsupported& e) {
    if (S->handle) { api.destroy(S->handle); S->handle = nullptr; }
    S->error_kind = "unsupported"; S->error = e.what();
    return S;
  } catch (const std::exception& e) {
    if (S->handle) { api.destroy(S->handle); S->handle = nullptr; }
    S->error_kind = "invalid"; S->error = e.what();
    return S;
  }
}

extern "C" const char* ksched_error(void* session) {
  Session* S = (Session*)session;
  return (S && !S->error_kind.empty()) ? S->error.c_str() : nullptr;
}
extern "C" const char* ksched_error_kind(void* session) {
  Session* S = (Session*)session;
  return (S && !S->error_kind.empty()) ? S->error_kind.c_str() : nullptr;
}
// The ctx deadline of Solve (scheduler.go:477-480): callable from another thread while ksched_solve runs; the solve
// returns what it has placed so far with timedOut set.
extern "C" int ksched_cancel(void* session) {
  Session* S = (Session*)session;
  if (!S || !S->handle || !S->api.cancel) return -1;
  return (int)S->api.cancel(S->handle);
}
// One probe of a resident cluster (ksolve_probe_create): `probe_json` = {"removeNodes": [node names], "pods": [uids]}.
// The new session shares the base session's tables on the host and on the device; close it before the base session.
// the session's node -> bound pods table (CSR) and the pods every simulation schedules (pending pods, pods of deleting nodes): built
// on the first sweep / the first probe that asks for a node's pods
static void ensure_sweep_tables(Session* B) {
  if (B->sweep_tables) return;
  const int ne = (int)B->node_names.size();
  B->node_pod_off.assign((size_t)ne + 1, 0);
  for (int p = 0; p < B->n_pods; ++p) { if (B->pod_node[p] >= 0) B->node_pod_off[(size_t)B->pod_node[p] + 1]++; else B->always_pods.push_back((uint32_t)p); }
  for (int e = 0; e < ne; ++e) B->node_pod_off[(size_t)e + 1] += B->node_pod_off[e];
  B->node_pod_list.assign(B->node_pod_off[ne], 0);
  std::vector<uint32_t> fill(B->node_pod_off.begin(), B->node_pod_off.end() - 1);
  for (int p = 0; p < B->n_pods; ++p) if (B->pod_node[p] >= 0) B->node_pod_list[fill[B->pod_node[p]]++] = (uint32_t)p;
  if (B->node_index.empty()) { B->node_index.reserve(B->node_names.size() * 2); for (size_t e = 0; e < B->node_names.size(); ++e) B->node_index[B->node_names[e]] = (int)e; }
  B->sweep_tables = true;
}
extern "C" void* ksched_probe(void* base_session, const char* probe_json) {
  Session* B = (Session*)base_session;
  Session* S = new Session();
  if (!B || !B->handle || B->base) { S->error_kind = "invalid"; S->error = "probe needs an open base session"; return S; }
  S->api = B->api;
  S->base = B;
  try {
    auto create = (decltype(&ksolve_probe_create))dlsym(B->api.lib, "ksolve_probe_create");
    if (!create) { S->error_kind = "load"; S->error = "solver library lacks ksolve_probe_create"; return S; }
    if (B->node_index.empty()) { B->node_index.reserve(B->node_names.size() * 2); for (size_t e = 0; e < B->node_names.size(); ++e) B->node_index[B->node_names[e]] = (int)e; }
    if (B->pod_index.empty()) for (int p = 0; p < B->n_pods; ++p) if (!B->uid_text[p].empty()) B->pod_index[B->uid_text[p]] = p;
    Value doc = kj::Parser(probe_json).parse();
    const size_t ne = B->node_names.size();
    const int nr1 = B->n_res + 1;
    std::vector<uint64_t> removed((ne + 63) / 64 + 1, 0);
    S->probe_removed.assign(ne, 0);
    std::vector<int64_t> lim = B->tmpl_lim;
    for (auto& n : doc.at("removeNodes").items()) {
      auto f = B->node_index.find(n.s());
      if (f == B->node_index.end()) throw std::runtime_error("probe removes an unknown node " + n.s());
      const int e = f->second;
      if (S->probe_removed[e]) continue;
      S->probe_removed[e] = 1;
      removed[e / 64] |= 1ull << (e % 64);
      const int t = B->node_tmpl[e];   // the pool gets the node's capacity back (scheduler.go:835-842)
      if (t >= 0) for (int r = 0; r < nr1; ++r) lim[(size_t)t * nr1 + r] += B->node_limit_cap[(size_t)e * nr1 + r];
    }
    std::vector<uint32_t> pods;
    S->probe_member.assign(B->n_pods, 0);
    if (doc.at("podsOfRemovedNodes").boolean_or(false)) {
      // the simulation a sweep runs for these candidates (helpers.go:53-155): the pods every simulation schedules, then the pods bound
      // to each removed node — a resident cluster whose pods travel as groups has no uid text to list them by
      if (B->pod_node.empty()) throw std::runtime_error("podsOfRemovedNodes needs a resident cluster (options.residentCluster)");
      ensure_sweep_tables(B);
      auto take = [&](uint32_t p) { if (!S->probe_member[p]) { S->probe_member[p] = 1; pods.push_back(p); } };
      for (uint32_t p : B->always_pods) take(p);
      for (auto& n : doc.at("removeNodes").items()) {
        const int e = B->node_index.find(n.s())->second;
        for (uint32_t i = B->node_pod_off[e]; i < B->node_pod_off[e + 1]; ++i) take(B->node_pod_list[i]);
      }
    }
    for (auto& u : doc.at("pods").items()) {
      auto f = B->pod_index.find(u.s());
      if (f == B->pod_index.end()) throw std::runtime_error("probe schedules an unknown pod " + u.s());
      if (S->probe_member[f->second]) throw std::runtime_error("probe lists pod " + u.s() + " twice");
      S->probe_member[f->second] = 1;
      pods.push_back((uint32_t)f->second);
    }
    S->probe_pods = (int)pods.size();
    ksolve_probe pr{};
    pr.removed_nodes = removed.data(); pr.n_pods = (uint32_t)pods.size(); pr.pods = pods.data();
    pr.tmpl_limits = lim.empty() ? nullptr : lim.data();
    ksolve_status st = create(B->handle, &pr, &S->handle);
    if (st != KSOLVE_OK) {
      S->error = S->handle ? S->api.last_error(S->handle) : "ksolve_probe_create failed";
      S->error_kind = st == KSOLVE_ERR_UNSUPPORTED ? "unsupported" : "create";
      if (S->handle) { S->api.destroy(S->handle); S->handle = nullptr; }
    }
  } catch (const std::exception& e) {
    if (S->handle) { S->api.destroy(S->handle); S->handle = nullptr; }
    S->error_kind = "invalid"; S->error = e.what();
  }
  return S;
}
// A whole consolidation sweep (ksolve_sweep): `sweep_json` = {"candidates": [[node, ...], ...], "prices": [sum of the candidates'
// prices per probe], "allSpot": [every candidate of the probe is a spot node], "detail": false}; a node is its name or its
// position in the problem's stateNodes list. Every probe is SimulateScheduling (disruption/helpers.go:53-155) of the resident
// cluster without its candidates — their pods, the pending pods and the pods of deleting nodes are scheduled — followed by
// computeConsolidation's verdict (consolidation.go:159-256): delete when no NodeClaim is needed, replace when exactly one is
// and a cheaper instance type remains after the price filter (nodeclaim.go:411-420), otherwise nothing. The probe descriptors
// are built here (CSR arrays over the session's node -> pods tables), not in the caller's language. Returns one document.
// the binary form of a sweep: candidates in as one CSR of positions in the stateNodes list, verdicts out as arrays (what a cgo caller
// hands over and reads back; no JSON on either side)
struct SweepArrays {
  uint32_t n; const uint32_t* cand_off; const uint32_t* cand_nodes; int multi_node;
  int32_t* decisions; uint8_t* all_scheduled; uint32_t* live_claims; int32_t* status; uint64_t* ref_evals;   // [n] each
  uint32_t* repl_off; uint32_t* repl_its; uint32_t repl_cap; uint8_t* repl_spot;                            // replace verdicts: instance type indices (CSR over probes), pinned-to-spot flag
  double* timings; uint32_t n_timings;
};
static char* sweep_impl(Session* B, const std::vector<ksolve_handle*>& replicas, const char* sweep_json, SweepArrays* A = nullptr);
extern "C" char* ksched_sweep(void* base_session, const char* sweep_json) {
  Session* B = (Session*)base_session;
  if (!B || !B->handle || B->base) return error_json("invalid", "sweep needs an open base session");
  return sweep_impl(B, {}, sweep_json);
}
// The same sweep spread over several devices: `sessions` were opened from ONE cluster document with different options.device
// (replicas); the first one supplies the descriptor / verdict tables, ksolve_sweep_replicas deals the probes out.
extern "C" char* ksched_sweep_replicas(void** sessions, int n, const char* sweep_json) {
  if (!sessions || n < 1) return error_json("invalid", "sweep needs at least one open base session");
  std::vector<ksolve_handle*> hs;
  for (int i = 0; i < n; ++i) {
    Session* S = (Session*)sessions[i];
    if (!S || !S->handle || S->base) return error_json("invalid", "sweep needs open base sessions");
    hs.push_back(S->handle);
  }
  return sweep_impl((Session*)sessions[0], hs, sweep_json);
}
// The sweep without JSON on either side — the form a cgo caller uses: probe p removes the stateNodes at positions
// cand_nodes[cand_off[p] .. cand_off[p + 1]) (their prices and capacity types come from the session's node table); out come
// computeConsolidation's verdict per probe (0 nothing, 1 delete, 2 replace, 3 spot-to-spot: the caller's per-probe path), whether all
// non-pending pods were scheduled, the live NodeClaims, the solver status, the reference-equivalent bin evaluations, and for replace
// verdicts the instance types that passed the price filter (indices into the problem's instanceTypes, CSR repl_off / repl_its) with the
// pinned-to-spot flag. timings[]: descriptors_ms, sweep_ms, verdicts_ms, upload_us, pack_us, finalize_us, download_us, pods, bin
// evaluations, node evaluations, node block steps, classes, it_words, nodes, node_dead0_us, req_words, resources, devices.
// Returns NULL, or an error document (ksched_free).
extern "C" char* ksched_sweep_arrays(void** sessions, int n_sessions, uint32_t n, const uint32_t* cand_off, const uint32_t* cand_nodes, int multi_node,
                                     int32_t* decisions, uint8_t* all_scheduled, uint32_t* live_claims, int32_t* status, uint64_t* ref_evals,
                                     uint32_t* repl_off, uint32_t* repl_its, uint32_t repl_cap, uint8_t* repl_spot, double* timings, uint32_t n_timings) {
  if (!sessions || n_sessions < 1 || !cand_off || (cand_off[n] && !cand_nodes)) return error_json("invalid", "sweep needs an open base session and candidate arrays");
  std::vector<ksolve_handle*> hs;
  for (int i = 0; i < n_sessions; ++i) {
    Session* S = (Session*)sessions[i];
    if (!S || !S->handle || S->base) return error_json("invalid", "sweep needs open base sessions");
    hs.push_back(S->handle);
  }
  if (hs.size() == 1) hs.clear();
  SweepArrays A{n, cand_off, cand_nodes, multi_node, decisions, all_scheduled, live_claims, status, ref_evals, repl_off, repl_its, repl_cap, repl_spot, timings, n_timings};
  return sweep_impl((Session*)sessions[0], hs, nullptr, &A);
}
static char* sweep_impl(Session* B, const std::vector<ksolve_handle*>& replicas, const char* sweep_json, SweepArrays* A) {
  try {
    auto run = (decltype(&ksolve_sweep))dlsym(B->api.lib, "ksolve_sweep");
    auto run_replicas = (decltype(&ksolve_sweep_replicas))dlsym(B->api.lib, "ksolve_sweep_replicas");
    if (replicas.size() > 1 && !run_replicas) return error_json("load", "solver library lacks ksolve_sweep_replicas");
    auto release = (decltype(&ksolve_sweep_results_free))dlsym(B->api.lib, "ksolve_sweep_results_free");
    if (!run || !release) return error_json("load", "solver library lacks ksolve_sweep");
    const auto t_begin = std::chrono::steady_clock::now();
    const int ne = (int)B->node_names.size(), n_res = B->n_res, nr1 = n_res + 1, T = B->n_templates, n_its = B->n_its;
    ensure_sweep_tables(B);
    Value doc = A ? Value::object() : kj::Parser(sweep_json).parse();
    // candidates: one list of nodes per simulation ("candidates": [[name | position, ...], ...]) or the same as CSR arrays of
    // positions in the stateNodes list ("candidateOff": n + 1 offsets, "candidateNodes")
    std::vector<uint32_t> node_off(1, 0), nodes;
    auto by_position = [&](int64_t i) {
      if (i < 0 || i >= (int64_t)B->node_input_index.size() || B->node_input_index[(size_t)i] < 0) throw std::runtime_error("sweep removes an unknown node index");
      return (uint32_t)B->node_input_index[(size_t)i];
    };
    // (a stamp per node instead of a scan over the probe's earlier candidates: a multi-node sweep names up to 101 nodes per probe)
    std::vector<uint32_t> seen_in((size_t)std::max(1, ne), 0);
    uint32_t probe_tag = 0;
    auto end_probe = [&](size_t n0) {   // a candidate named twice is one candidate; the order given stays (the prices are summed in it)
      size_t w = n0;
      ++probe_tag;
      for (size_t r = n0; r < nodes.size(); ++r) {
        const uint32_t e = nodes[r];
        if (e < seen_in.size() && seen_in[e] == probe_tag) continue;
        if (e < seen_in.size()) seen_in[e] = probe_tag;
        nodes[w++] = e;
      }
      nodes.resize(w);
      node_off.push_back((uint32_t)nodes.size());
    };
    if (A) {
      nodes.reserve(A->cand_off[A->n]);
      for (uint32_t p = 0; p < A->n; ++p) {
        if (A->cand_off[p + 1] < A->cand_off[p]) throw std::runtime_error("sweep: candidate offsets must not decrease");
        const size_t n0 = nodes.size();
        for (uint32_t j = A->cand_off[p]; j < A->cand_off[p + 1]; ++j) nodes.push_back(by_position((int64_t)A->cand_nodes[j]));
        end_probe(n0);
      }
    } else if (doc.has("candidateOff")) {
      const auto& off = doc.at("candidateOff").items();
      const auto& flat = doc.at("candidateNodes").items();
      if (off.empty() || off.back().i(-1) != (int64_t)flat.size()) throw std::runtime_error("sweep: candidateOff does not cover candidateNodes");
      nodes.reserve(flat.size());
      for (size_t p = 0; p + 1 < off.size(); ++p) {
        const int64_t a = off[p].i(-1), e = off[p + 1].i(-1);
        if (a < 0 || e < a || e > (int64_t)flat.size()) throw std::runtime_error("sweep: candidateOff must not decrease");
        const size_t n0 = nodes.size();
        for (int64_t j = a; j < e; ++j) nodes.push_back(by_position(flat[(size_t)j].i(-1)));
        end_probe(n0);
      }
    } else {
      for (auto& cv : doc.at("candidates").items()) {
        const size_t n0 = nodes.size();
        for (auto& nv : cv.items()) {
          if (nv.kind == Value::Str) { auto f = B->node_index.find(nv.s()); if (f == B->node_index.end()) throw std::runtime_error("sweep removes an unknown node " + nv.s()); nodes.push_back((uint32_t)f->second); }
          else nodes.push_back(by_position(nv.i(-1)));
        }
        end_probe(n0);
      }
    }
    const uint32_t n = (uint32_t)node_off.size() - 1;
    // "prices" / "allSpot" may be left out: the session then takes every candidate's price from its own offering (node_price) and
    // its capacity type from its label. "multiNode": the simulations are prefixes of MultiNodeConsolidation's binary search —
    // a replace verdict over several candidates goes through filterOutSameInstanceType (multinodeconsolidation.go:209-246)
    const bool own_prices = !doc.has("prices");
    static const std::vector<Value> no_items;
    const auto& prices = own_prices ? no_items : doc.at("prices").items();
    const auto& all_spot = own_prices ? no_items : doc.at("allSpot").items();
    const bool multi_node = A ? A->multi_node != 0 : doc.at("multiNode").boolean_or(false);
    const bool detail = doc.at("detail").boolean_or(false);
    const bool spot_to_spot = B->root.at("options").at("spotToSpotConsolidation").boolean_or(false);
    // the displaced pods of every probe: sizes first, then every probe fills its own slice (independent: on a few threads)
    std::vector<uint32_t> pod_off(n + 1, 0), pods;
    const size_t n_always = B->always_pods.size();
    for (uint32_t p = 0; p < n; ++p) {
      size_t m = n_always;
      for (uint32_t j = node_off[p]; j < node_off[p + 1]; ++j) m += B->node_pod_off[nodes[j] + 1] - B->node_pod_off[nodes[j]];
      pod_off[p + 1] = pod_off[p] + (uint32_t)m;
    }
    pods.resize(pod_off[n]);
    std::vector<int64_t> lims;
    const bool limits = !B->tmpl_lim.empty();
    if (limits) lims.resize((size_t)n * T * nr1);
    parallel_for(n, [&](size_t p) {
      uint32_t* dst = pods.data() + pod_off[p];
      if (n_always) { memcpy(dst, B->always_pods.data(), n_always * 4); dst += n_always; }
      for (uint32_t j = node_off[p]; j < node_off[p + 1]; ++j) {
        const uint32_t a = B->node_pod_off[nodes[j]], e = B->node_pod_off[nodes[j] + 1];
        if (e > a) { memcpy(dst, B->node_pod_list.data() + a, (size_t)(e - a) * 4); dst += e - a; }
      }
      if (limits) {
        int64_t* l = lims.data() + p * T * nr1;
        memcpy(l, B->tmpl_lim.data(), (size_t)T * nr1 * 8);
        for (uint32_t j = node_off[p]; j < node_off[p + 1]; ++j) {   // the pool gets the node's capacity back (scheduler.go:835-842)
          const int e = (int)nodes[j], t = B->node_tmpl[e];
          if (t >= 0) for (int r = 0; r < nr1; ++r) l[(size_t)t * nr1 + r] += B->node_limit_cap[(size_t)e * nr1 + r];
        }
      }
    });
    ksolve_sweep_desc sd{};
    sd.n_probes = n; sd.node_off = node_off.data(); sd.nodes = nodes.data(); sd.pod_off = pod_off.data(); sd.pods = pods.data();
    sd.tmpl_limits = limits ? lims.data() : nullptr;
    const auto t_desc = std::chrono::steady_clock::now();
    ksolve_sweep_results res{};
    ksolve_status st = replicas.size() > 1 ? run_replicas(const_cast<ksolve_handle**>(replicas.data()), (uint32_t)replicas.size(), &sd, &res) : run(B->handle, &sd, &res);
    const auto t_solved = std::chrono::steady_clock::now();
    if (st != KSOLVE_OK) return error_json(st == KSOLVE_ERR_UNSUPPORTED ? "unsupported" : st == KSOLVE_ERR_CAPACITY ? "capacity" : "solve", B->api.last_error(B->handle));

    // ---- verdicts ----
    Flattener& fl = B->fl;
    Dictionary& D = fl.dict;
    const ks::Dict& kd = fl.kd;
    const int nk = kd.n_keys;
    const ksolve_claims& cl = res.claims;
    auto ct_index = [&](const char* name) { if (kd.key_ct < 0) return -1; auto f = D.value_index[kd.key_ct].find(name); return f == D.value_index[kd.key_ct].end() ? -1 : f->second; };
    const int ct_order[3] = {ct_index("reserved"), ct_index("spot"), ct_index("on-demand")};
    Value decisions = Value::array(), scheduled = Value::array(), n_claims_j = Value::array(), status_j = Value::array(), refs = Value::array(), repl = Value::array(), reasons = Value::object();
    Value details = Value::array();
    auto uid_of = [&](int p) { if (!B->uid_text[p].empty()) return B->uid_text[p]; std::string u; uint64_t a, b; group_uid(B->group_of_pod[p].first, B->group_of_pod[p].second, a, b, &u); return u; };
    // every probe's verdict is computed on its own (a few threads), the document is put together afterwards in probe order
    struct Verdict { int decision = 0; bool all_ok = true; uint32_t live = 0; std::vector<int> cheaper; bool pin_spot = false; const char* reason = nullptr; };
    std::vector<Verdict> verdicts(n);
    parallel_for(n, [&](size_t pp) {
      const uint32_t p = (uint32_t)pp;
      Verdict& V = verdicts[p];
      int decision = 0;   // 0 no-op, 1 delete, 2 replace
      bool all_ok = true;
      const uint32_t c0 = res.claim_off[p], C = res.claim_off[p + 1] - c0;
      if (res.status[p] != KSOLVE_OK) all_ok = false;
      for (uint32_t i = pod_off[p]; i < pod_off[p + 1] && all_ok; ++i) {
        const int g = (int)pods[i];
        if (B->pod_pending_flag[g]) continue;                         // AllNonPendingPodsScheduled (scheduler.go:388-392)
        const int a = res.pod_assignment[i];
        if (a == -1) all_ok = false;
        else if (a >= 0 && cl.truncation_failed && cl.truncation_failed[c0 + (uint32_t)a]) all_ok = false;   // the claim is dropped, its pods fail (scheduler.go:426-431)
        else if (a <= -2 && !B->node_initialized[(size_t)(-2 - a)] && !B->pod_deleting_flag[g]) all_ok = false;   // UninitializedNodeError (helpers.go:133-153)
      }
      uint32_t live = 0, only = 0;
      for (uint32_t c = 0; c < C; ++c) if (!(cl.truncation_failed && cl.truncation_failed[c0 + c])) { live++; only = c0 + c; }
      if (all_ok && live == 0) decision = 1;
      else if (all_ok && live == 1) {
        const uint32_t c = only;
        double price = p < prices.size() ? prices[p].d(0) : 0.0;
        bool every_spot = p < all_spot.size() && all_spot[p].boolean_or(false);
        if (own_prices) {   // getCandidatePrices (consolidation.go:345-356) and the allSpot test of :215-222, from the session's node table
          price = 0.0; every_spot = node_off[p + 1] > node_off[p];
          for (uint32_t j = node_off[p]; j < node_off[p + 1]; ++j) { price += B->node_price[nodes[j]]; every_spot = every_spot && B->node_spot[nodes[j]]; }
        }
        ks::ReqRef r;
        r.mask = cl.req_mask + (size_t)c * cl.req_words; r.defined = cl.req_defined[c]; r.complement = cl.req_complement[c];
        r.has_gte = cl.req_has_gte[c]; r.has_lte = cl.req_has_lte[c]; r.gte = cl.req_gte + (size_t)c * nk; r.lte = cl.req_lte + (size_t)c * nk; r.minv = nullptr;
        auto has = [&](int key, int v) { return key < 0 || !ks::bit(r.defined, key) || ks::req_has(kd, r, key, kd.key_word_off[key] + (uint32_t)(v >> 6), v & 63); };
        const bool ct_defined = kd.key_ct >= 0 && ks::bit(r.defined, kd.key_ct);
        const bool spot_ok = !ct_defined || (ct_order[1] >= 0 && has(kd.key_ct, ct_order[1]));
        const bool od_ok = !ct_defined || (ct_order[2] >= 0 && has(kd.key_ct, ct_order[2]));
        if (every_spot && spot_ok) {
          // computeSpotToSpotConsolidation (consolidation.go:261-342): behind its feature gate; with the gate on the caller takes
          // the per-probe path (the 15-cheapest rule needs the whole ordered list)
          if (spot_to_spot) { decision = 3; }
        } else {
          // Offerings.Available().WorstLaunchPrice(reqs) (types.go:587-598): reserved, then spot, then on-demand
          // pin_spot: the claim's requirements after consolidation.go:238-243 narrowed capacity-type to In [spot]
          bool pin_spot = false;
          auto worst = [&](int it) {
            for (int q = 0; q < 3; ++q) {
              if (ct_order[q] < 0) continue;
              double mx = -1; bool any = false;
              for (auto& o : B->it_offerings[it]) {
                if (!o.available || o.ct != ct_order[q] || !has(kd.key_zone, o.zone) || !has(kd.key_ct, o.ct)) continue;
                if (pin_spot && o.ct != ct_order[1]) continue;
                if (o.rid >= 0 && !has(B->k_rid, o.rid)) continue;
                if (!any || o.price > mx) mx = o.price;
                any = true;
              }
              if (any) return mx;
            }
            return 1.0 / 0.0;
          };
          std::vector<int> cheaper;
          if (cl.ordered_instance_types) { for (uint32_t i = 0; i < cl.ordered_count[c]; ++i) { const int it = cl.ordered_instance_types[(size_t)c * cl.n_instance_types + i]; if (worst(it) < price) cheaper.push_back(it); } }
          else for (int it = 0; it < n_its; ++it) if (((cl.it_mask[(size_t)c * cl.it_words + it / 64] >> (it % 64)) & 1) && worst(it) < price) cheaper.push_back(it);
          auto min_values_ok = [&](const std::vector<int>& its_) {
            for (int k = 0; k < nk; ++k) {
              const int want = cl.req_min_values[(size_t)c * nk + k];
              if (want < 0 || !ks::bit(r.defined, k)) continue;
              std::set<std::string> seen;
              for (int it : its_) for (auto& e : B->it_exprs[it]) if (e.key == D.keys[k]) seen.insert(e.values.begin(), e.values.end());
              if ((int)seen.size() < want) return false;
            }
            return true;
          };
          // InstanceTypes.SatisfiesMinValues after the price filter (nodeclaim.go:416-418, types.go:399-433); an empty list
          // satisfies it (no error), so "no cheaper type" is reported as that, not as a minValues failure (consolidation.go:228-233)
          bool mv_ok = cheaper.empty() || min_values_ok(cheaper);
          if (mv_ok && multi_node && !cheaper.empty() && node_off[p + 1] - node_off[p] > 1) {
            // filterOutSameInstanceType (multinodeconsolidation.go:209-246): when a replacement option is one of the types being
            // removed, the replacement must be cheaper than the cheapest candidate of that type — else deleting the others is the
            // better command; RemoveInstanceTypeOptionsByPriceAndMinValues again with that price
            std::map<int, double> existing;
            for (uint32_t j = node_off[p]; j < node_off[p + 1]; ++j) {
              const int it = B->node_it[nodes[j]];
              if (it < 0) continue;
              auto f = existing.find(it);
              if (f == existing.end()) existing[it] = B->node_price[nodes[j]]; else if (B->node_price[nodes[j]] < f->second) f->second = B->node_price[nodes[j]];
            }
            double max_price = 1.0 / 0.0;
            for (int it : cheaper) { auto f = existing.find(it); if (f != existing.end() && f->second < max_price) max_price = f->second; }
            // the Replacement wraps the NodeClaim computeConsolidation just narrowed (disruption/types.go:224-226): when it was
            // pinned to spot, the options are priced by their spot offerings here — one without any has WorstLaunchPrice
            // MaxFloat64 and fails the strict '<' even against a MaxFloat64 ceiling (found by oracle/consolidation.hpp, round 4)
            pin_spot = !ct_defined || (spot_ok && od_ok);
            std::vector<int> kept;
            for (int it : cheaper) if (worst(it) < max_price) kept.push_back(it);
            cheaper.swap(kept);
            if (cheaper.empty()) V.reason = "every replacement option is one of the types being removed, or more expensive";
            else mv_ok = min_values_ok(cheaper);
          }
          if (!mv_ok) V.reason = "minValues requirement is not met after filtering by price";
          else if (cheaper.empty() && !V.reason) V.reason = "Can't replace with a cheaper node";
          else if (!cheaper.empty()) {
            decision = 2;
            V.cheaper.swap(cheaper);
            V.pin_spot = !ct_defined || (spot_ok && od_ok);   // consolidation.go:238-243
          }
        }
      }
      V.decision = decision; V.all_ok = all_ok; V.live = live;
    });
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (A) {
      uint32_t at = 0;
      bool overflow = false;
      for (uint32_t p = 0; p < n; ++p) {
        const Verdict& V = verdicts[p];
        if (A->decisions) A->decisions[p] = V.decision;
        if (A->all_scheduled) A->all_scheduled[p] = V.all_ok ? 1 : 0;
        if (A->live_claims) A->live_claims[p] = V.live;
        if (A->status) A->status[p] = (int32_t)res.status[p];
        if (A->ref_evals) A->ref_evals[p] = res.ref_bin_evaluations[p];
        if (A->repl_spot) A->repl_spot[p] = V.pin_spot ? 1 : 0;
        if (A->repl_off) A->repl_off[p] = at;
        if (V.decision == 2) {
          if (A->repl_its && at + V.cheaper.size() <= A->repl_cap) memcpy(A->repl_its + at, V.cheaper.data(), V.cheaper.size() * 4);
          else if (A->repl_its) overflow = true;
          at += (uint32_t)V.cheaper.size();
        }
      }
      if (A->repl_off) A->repl_off[n] = at;
      const auto t_end = std::chrono::steady_clock::now();
      const double tv[] = {ms(t_begin, t_desc), ms(t_desc, t_solved), ms(t_solved, t_end), res.us_upload, res.us_pack, res.us_finalize, res.us_download, (double)pods.size(),
                           (double)res.total_bin_evaluations, (double)res.total_node_evaluations, (double)res.total_node_block_steps, (double)res.n_classes, (double)res.it_words, (double)res.n_nodes,
                           res.us_node_dead0, (double)kd.req_words, (double)n_res, (double)std::max<size_t>(1, replicas.size())};
      for (uint32_t i = 0; i < A->n_timings && i < sizeof(tv) / sizeof(tv[0]); ++i) A->timings[i] = tv[i];
      release(&res);
      if (overflow) return error_json("capacity", "replacement instance types exceed repl_cap (repl_off[n] says how many there are)");
      return nullptr;
    }
    for (uint32_t p = 0; p < n; ++p) {
      const Verdict& V = verdicts[p];
      if (V.reason) reasons.add_new(std::to_string(p), Value::string(V.reason));
      if (V.decision == 2) {
        Value rj = Value::object();
        rj.set("probe", Value::integer(p));
        std::vector<std::string> names;
        for (int it : V.cheaper) names.push_back(B->it_names[it]);
        std::sort(names.begin(), names.end());
        Value nj = Value::array();
        for (auto& nm : names) nj.push(Value::string(nm));
        rj.set("instanceTypes", nj);
        rj.set("capacityType", V.pin_spot ? Value::string("spot") : Value());
        repl.push(rj);
      }
      decisions.push(Value::integer(V.decision)); scheduled.push(Value::boolean(V.all_ok)); n_claims_j.push(Value::integer(V.live));
      status_j.push(Value::integer(res.status[p])); refs.push(Value::integer((int64_t)res.ref_bin_evaluations[p]));
      if (detail) {
        // where every pod of the probe went: uid -> claim index (>= 0), node name, or null
        Value dj = Value::object(), where = Value::object();
        for (uint32_t i = pod_off[p]; i < pod_off[p + 1]; ++i) {
          const int a = res.pod_assignment[i];
          where.add_new(uid_of((int)pods[i]), a >= 0 ? Value::integer(a) : a <= -2 ? Value::string(B->node_names[(size_t)(-2 - a)]) : Value());
        }
        dj.set("pods", where);
        details.push(dj);
      }
    }
    const auto t_end = std::chrono::steady_clock::now();
    Value out = Value::object();
    out.set("decisions", decisions); out.set("allNonPendingPodsScheduled", scheduled); out.set("claims", n_claims_j); out.set("status", status_j);
    out.set("referenceBinEvaluations", refs); out.set("replacements", repl); out.set("reasons", reasons);
    if (detail) out.set("details", details);
    Value tj = Value::object();
    tj.set("descriptors_ms", Value::number(ms(t_begin, t_desc))); tj.set("sweep_ms", Value::number(ms(t_desc, t_solved))); tj.set("verdicts_ms", Value::number(ms(t_solved, t_end)));
    tj.set("upload_us", Value::number(res.us_upload)); tj.set("pack_us", Value::number(res.us_pack)); tj.set("finalize_us", Value::number(res.us_finalize)); tj.set("download_us", Value::number(res.us_download));
    tj.set("probes", Value::integer(n)); tj.set("pods", Value::integer((int64_t)pods.size())); tj.set("devices", Value::integer((int64_t)std::max<size_t>(1, replicas.size())));
    tj.set("bin_evaluations", Value::integer((int64_t)res.total_bin_evaluations)); tj.set("node_evaluations", Value::integer((int64_t)res.total_node_evaluations));
    tj.set("node_block_steps", Value::integer((int64_t)res.total_node_block_steps));
    tj.set("req_words", Value::integer(kd.req_words)); tj.set("resources", Value::integer(n_res)); tj.set("classes", Value::integer((int64_t)res.n_classes));
    tj.set("it_words", Value::integer((int64_t)res.it_words)); tj.set("nodes", Value::integer((int64_t)res.n_nodes)); tj.set("node_dead0_us", Value::number(res.us_node_dead0));
    out.set("timings", tj);
    release(&res);
    return dup_json(out);
  } catch (const std::exception& e) {
    return error_json("invalid", e.what());
  }
}
// The flat per-pod outputs of the session's last Solve(want_results = 2): pod i (position in the problem: the explicit pods, then
// the groups' pods in order) sits in slot[i] of NodeClaim assign[i] (>= 0: index into newNodeClaims), of existing node -2 - assign[i],
// or nowhere (-1). Returns the number of pods (copies min(capacity, pods) entries).
extern "C" uint32_t ksched_assignment(void* session, int32_t* assign, uint32_t* slot, uint32_t capacity) {
  Session* S = (Session*)session;
  if (!S) return 0;
  const uint32_t n = (uint32_t)S->last_assign.size(), m = std::min(n, capacity);
  if (assign && m) memcpy(assign, S->last_assign.data(), (size_t)m * 4);
  if (slot && m) memcpy(slot, S->last_slot.data(), (size_t)m * 4);
  return n;
}
// NodeClaim.Pods of every new NodeClaim of the last Solve(want_results >= 2) as one CSR: pods[claim_off[c] .. claim_off[c + 1]) are
// the positions of claim c's pods in slot order (the order the reference appended them in, nodeclaim.go:248). A pod's slot IS its
// place inside its claim, so this is one counting pass and one scatter — no sort, no uid text. Returns the number of pods on
// NodeClaims (fills at most `capacity` of them; claim_off has n_claims + 1 entries).
extern "C" uint32_t ksched_pods_by_claim(void* session, uint32_t n_claims, uint32_t* claim_off, uint32_t* pods, uint32_t capacity) {
  Session* S = (Session*)session;
  if (!S || !claim_off) return 0;
  const size_t n = S->last_assign.size();
  for (uint32_t c = 0; c <= n_claims; ++c) claim_off[c] = 0;
  for (size_t p = 0; p < n; ++p) { const int32_t a = S->last_assign[p]; if (a >= 0 && (uint32_t)a < n_claims) claim_off[(size_t)a + 1]++; }
  for (uint32_t c = 0; c < n_claims; ++c) claim_off[c + 1] += claim_off[c];
  const uint32_t total = claim_off[n_claims];
  if (pods && total <= capacity)
    for (size_t p = 0; p < n; ++p) {
      const int32_t a = S->last_assign[p];
      if (a < 0 || (uint32_t)a >= n_claims) continue;
      const uint32_t at = claim_off[a] + S->last_slot[p];
      if (at < claim_off[(size_t)a + 1]) pods[at] = (uint32_t)p;
    }
  return total;
}
extern "C" void ksched_close(void* session) {
  Session* S = (Session*)session;
  if (!S) return;
  if (S->handle) S->api.destroy(S->handle);
  delete S;
}

// Solve(): one pass of the hot path on the device with the inputs already resident in HBM. want_results = 0 skips
// the rehydration of NodeClaims (counters, timings and packing cost are always returned).
// Results of one finished device solve as a JSON document (takes ownership of `res`).
static char* results_json(Session* S, ksolve_results& res, ksolve_status st, int want_results) {
  Api& api = S->api;
  ksolve_handle* handle = S->handle;
  Session* own = S;                       // the session whose handle ran (a probe owns only that and its membership tables)
  if (S->base) S = S->base;               // names, dictionaries and scales are the base session's
  auto in_probe = [&](int p) { return !own->base || own->probe_member[p]; };
  auto node_there = [&](size_t e) { return !own->base || !own->probe_removed[e]; };
  Flattener& fl = S->fl;
  Dictionary& D = fl.dict;
  const int n_pods = S->n_pods, n_rows = S->n_rows, n_its = S->n_its, n_res = S->n_res, it_words = S->it_words;
  const int nk = fl.kd.n_keys, rw = fl.kd.req_words;
  const UidTexts& uid_text = S->uid_text;
  const std::vector<std::pair<uint64_t, uint64_t>>& group_of_pod = S->group_of_pod;
  const std::vector<std::string>& res_names = S->res_names;
  const std::vector<i128>& scale = S->scale;
  try {
    Value timings = Value::array();
    {
      if (st != KSOLVE_OK && st != KSOLVE_ERR_CANCELLED) {
        std::string msg = api.last_error(handle);
        return error_json(st == KSOLVE_ERR_UNSUPPORTED ? "unsupported" : st == KSOLVE_ERR_CAPACITY ? "capacity" : "solve", msg);
      }
      Value t = Value::object();
      t.set("upload_us", Value::number(res.us_upload)); t.set("prepass_us", Value::number(res.us_prepass)); t.set("pack_us", Value::number(res.us_pack));
      t.set("finalize_us", Value::number(res.us_finalize)); t.set("download_us", Value::number(res.us_download));
      if (api.kernel_ms) {
        t.set("pack_kernel_ms", Value::number(api.kernel_ms(handle, "ksolve_pack")));
        t.set("classify_ms", Value::number(api.kernel_ms(handle, "classify")));
        t.set("row_hash_ms", Value::number(api.kernel_ms(handle, "row_hash")));
        t.set("sort_ms", Value::number(api.kernel_ms(handle, "sort")));
        t.set("it_index_ms", Value::number(api.kernel_ms(handle, "it_index")));
      }
      timings.push(t);
    }

    // ---- rehydrate Results ----
    Value out = Value::object();
    auto uid_of = [&](int p) { if (!uid_text[p].empty()) return uid_text[p]; std::string s; uint64_t a, b; group_uid(group_of_pod[p].first, group_of_pod[p].second, a, b, &s); return s; };
    const ksolve_claims& cl = res.claims;
    Value counters = Value::object();
    counters.set("binEvaluations", Value::integer((int64_t)res.bin_evaluations));
    counters.set("instanceTypeEvaluations", Value::integer((int64_t)res.it_evaluations));
    counters.set("referenceBinEvaluations", Value::integer((int64_t)res.ref_bin_evaluations));
    counters.set("pops", Value::integer((int64_t)res.queue_pops)); counters.set("sorts", Value::integer((int64_t)res.sorts));
    counters.set("slowSorts", Value::integer((int64_t)res.slow_sorts)); counters.set("relaxations", Value::integer((int64_t)res.relaxations));
    counters.set("pods", Value::integer(own->base ? own->probe_pods : n_pods)); counters.set("claims", Value::integer(cl.n_claims));
    counters.set("engine", Value::string(res.engine_used == 2 ? "cursor" : res.engine_used == 3 ? "spread" : "general")); counters.set("cursorClaimStateInHBM", Value::boolean(res.engine_used == 2 && res.cursor_wide != 0)); counters.set("cursorMemoryPlan", Value::integer(res.engine_used == 2 ? (int64_t)res.cursor_wide : -1)); counters.set("cursorAttempts", Value::integer((int64_t)res.cursor_attempts)); counters.set("engineFallbackReason", Value::integer((int64_t)res.engine_fallback_reason));
    counters.set("rows", Value::integer(n_rows)); counters.set("instanceTypes", Value::integer(n_its)); counters.set("strictTableShared", Value::boolean(S->strict_shared));
    counters.set("topologyGroups", Value::integer(S->n_topo_groups)); counters.set("topologyAliasClasses", Value::integer(S->n_alias_classes));
    counters.set("reqWords", Value::integer(rw)); counters.set("itWords", Value::integer(it_words)); counters.set("keys", Value::integer(nk)); counters.set("resources", Value::integer(n_res));
    { Value pc = Value::array(); for (int i = 0; i < 24; ++i) pc.push(Value::integer((int64_t)res.phase_cycles[i])); counters.set("phaseCycles", pc); }
    out.set("counters", counters);
    out.set("timings", timings);
    out.set("timedOut", Value::boolean(st == KSOLVE_ERR_CANCELLED));
    out.set("packingCost", Value::number(res.packing_cost));
    {
      // the per-instance-type (NodeClaim count, $/h) vector of this packing (ksolve_packing_vector): sparse, [type, count, cost]
      auto pvec = (decltype(&ksolve_packing_vector))dlsym(api.lib, "ksolve_packing_vector");
      if (pvec && st == KSOLVE_OK) {
        std::vector<double> cnt(n_its), cst(n_its);
        if (pvec(handle, &res, cnt.data(), cst.data()) == KSOLVE_OK) {
          Value pv = Value::array();
          for (int i = 0; i < n_its; ++i) if (cnt[i] > 0) { Value e = Value::array(); e.push(Value::integer(i)); e.push(Value::number(cnt[i])); e.push(Value::number(cst[i])); pv.push(e); }
          out.set("packingVector", pv);
        }
      }
    }
    int unscheduled = 0;
    for (int p = 0; p < n_pods; ++p) if (in_probe(p) && res.pod_assignment[p] == -1) unscheduled++;
    { int64_t there = 0; for (size_t e = 0; e < S->node_names.size(); ++e) if (node_there(e)) there++; counters.set("existingNodes", Value::integer(there)); }
    out.set("scheduledPods", Value::integer((own->base ? own->probe_pods : n_pods) - unscheduled));
    if (want_results) {
      std::vector<std::vector<std::pair<uint32_t, int>>> members(want_results == 1 ? cl.n_claims : 0);
      std::vector<uint32_t> member_count(cl.n_claims, 0);
      std::vector<std::vector<std::pair<uint32_t, int>>> node_members(S->node_names.size());
      Value errs = Value::object();
      if (want_results >= 2) {
        // the claims without their pod lists: the per-pod outputs stay flat (ksched_assignment hands them over as two arrays —
        // a caller that holds its pods by position needs no uid text to put them on their NodeClaims)
        own->last_assign.assign(res.pod_assignment, res.pod_assignment + n_pods);
        own->last_slot.assign(res.pod_slot, res.pod_slot + n_pods);
        if (cl.truncation_failed) {
          // a NodeClaim that TruncateInstanceTypes drops (scheduler.go:426-431) is not in newNodeClaims: the indices ksched_assignment /
          // ksched_pods_by_claim hand out are positions in the EMITTED list, the dropped claims' pods are unscheduled
          std::vector<int32_t> emitted(cl.n_claims, -1);
          int32_t next = 0;
          bool any = false;
          for (uint32_t c = 0; c < cl.n_claims; ++c) { if (cl.truncation_failed[c]) any = true; else emitted[c] = next++; }
          if (any) for (auto& a : own->last_assign) if (a >= 0 && (uint32_t)a < cl.n_claims) a = emitted[(size_t)a];
        }
      }
      for (int p = 0; p < n_pods; ++p) {
        if (!in_probe(p)) continue;
        int a = res.pod_assignment[p];
        if (a >= 0) { member_count[a]++; if (want_results == 1) members[a].push_back({res.pod_slot[p], p}); }
        else if (a <= -2) node_members[-2 - a].push_back({res.pod_slot[p], p});
        else { Value e = Value::object(); e.set("code", Value::integer(res.pod_error[p])); e.set("diag", Value::integer(res.pod_error_diag[p])); errs.add_new(uid_of(p), e); }
      }
      Value claims = Value::array();
      for (uint32_t c = 0; c < cl.n_claims; ++c) {
        Value cj = Value::object();
        int t = cl.template_idx[c];
        cj.set("nodePool", Value::string(S->pool_names[t]));
        char hb[64];
        snprintf(hb, sizeof hb, "hostname-placeholder-%04u", cl.hostname_seq[c]);
        cj.set("hostname", Value::string(hb));
        Value pj = Value::array();
        if (want_results == 1) { std::sort(members[c].begin(), members[c].end()); for (auto& m : members[c]) pj.push(Value::string(uid_of(m.second))); }   // want_results 2: the claims without their pod lists
        cj.set("pods", pj);
        cj.set("podCount", Value::integer((int64_t)member_count[c]));
        Value itj = Value::array();
        // want_results 3: the options as positions in the problem's instanceTypes list (a caller that holds the catalogue needs no
        // names: 2,763 NodeClaims x a few hundred options are megabytes of repeated strings)
        auto option = [&](int i) { return want_results == 3 ? Value::integer(i) : Value::string(S->it_names[i]); };
        if (cl.ordered_instance_types) {   // Results.TruncateInstanceTypes: price order, capped (scheduler.go:419-437)
          for (uint32_t i = 0; i < cl.ordered_count[c]; ++i) itj.push(option((int)cl.ordered_instance_types[(size_t)c * cl.n_instance_types + i]));
        } else {
          for (int i = 0; i < n_its; ++i) if ((cl.it_mask[(size_t)c * cl.it_words + i / 64] >> (i % 64)) & 1) itj.push(option(i));
        }
        cj.set(want_results == 3 ? "instanceTypeIndices" : "instanceTypes", itj);
        if (cl.truncation_failed && cl.truncation_failed[c]) {
          // the claim is dropped and its pods fail with the minValues error (scheduler.go:426-431)
          auto failed = [&](int p) { Value e = Value::object(); e.set("code", Value::integer(KSOLVE_POD_MIN_VALUES)); e.set("diag", Value::integer(128)); errs.add_new(uid_of(p), e); };
          if (want_results == 1) for (auto& m : members[c]) failed(m.second);
          else for (int p = 0; p < n_pods; ++p) if (in_probe(p) && res.pod_assignment[p] == (int)c) failed(p);
          continue;
        }
        Value rj = Value::array();
        // requirements in key-name order, like the oracle's std::map
        std::vector<std::pair<std::string, int>> keys;
        for (int k = 0; k < nk; ++k) if ((cl.req_defined[c] >> k) & 1) keys.push_back({D.keys[k], k});
        std::sort(keys.begin(), keys.end());
        for (auto& kk : keys) {
          int k = kk.second;
          Value r = Value::object();
          r.set("key", Value::string(kk.first));
          bool comp = (cl.req_complement[c] >> k) & 1;
          r.set("complement", Value::boolean(comp));
          std::vector<std::string> vals;
          for (size_t v = 0; v < D.values[k].size(); ++v) { size_t pos = (size_t)fl.key_word_off[k] * 64 + v; if ((cl.req_mask[(size_t)c * cl.req_words + pos / 64] >> (pos % 64)) & 1) vals.push_back(D.values[k][v]); }
          std::sort(vals.begin(), vals.end());
          Value vj = Value::array();
          for (auto& v : vals) vj.push(Value::string(v));
          r.set("values", vj);
          r.set("gte", ((cl.req_has_gte[c] >> k) & 1) ? Value::integer(cl.req_gte[(size_t)c * nk + k]) : Value());
          r.set("lte", ((cl.req_has_lte[c] >> k) & 1) ? Value::integer(cl.req_lte[(size_t)c * nk + k]) : Value());
          int mv = cl.req_min_values[(size_t)c * nk + k];
          r.set("minValues", mv >= 0 ? Value::integer(mv) : Value());
          r.set("operator", Value::string(comp ? (vals.empty() ? "Exists" : "NotIn") : (vals.empty() ? "DoesNotExist" : "In")));
          rj.push(r);
        }
        cj.set("requirements", rj);
        Value qj = Value::object();
        for (int r = 0; r < n_res; ++r) {
          int64_t v = cl.requests[(size_t)c * n_res + r];
          if (v != 0 || res_names[r] == "pods") qj.set(res_names[r], Value::string(i128_str((i128)v * scale[r])));
        }
        cj.set("requests", qj);
        Value ann = Value::object();
        ann.set("karpenter.sh/nodeclaim-min-values-relaxed", Value::string(cl.min_values_relaxed[c] ? "true" : "false"));
        cj.set("annotations", ann);
        {
          Value ro = Value::array();
          if (cl.reserved_mask && S->k_rid >= 0) for (int i = 0; i < 64; ++i) if ((cl.reserved_mask[c] >> i) & 1) ro.push(Value::string(D.values[S->k_rid][i]));
          cj.set("reservedOfferings", ro);
        }
        cj.set("cheapestPrice", Value::number(cl.cheapest_price[c] < 1e300 ? cl.cheapest_price[c] : -1.0));
        claims.push(cj);
      }
      out.set("newNodeClaims", claims);
      Value ens = Value::array();
      for (size_t e = 0; e < S->node_names.size(); ++e) {
        if (!node_there(e)) continue;
        if (own->base && node_members[e].empty()) continue;   // a probe of a 10k-node cluster reports the nodes that took pods
        Value ej = Value::object();
        ej.set("name", Value::string(S->node_names[e]));
        std::sort(node_members[e].begin(), node_members[e].end());
        Value pj = Value::array();
        for (auto& m : node_members[e]) pj.push(Value::string(uid_of(m.second)));
        ej.set("pods", pj);
        ej.set("initialized", Value::boolean(S->node_initialized[e] != 0));
        ens.push(ej);
      }
      out.set("existingNodes", ens);
      out.set("podErrors", errs);
    }
    api.results_free(&res);
    return dup_json(out);
  } catch (const std::exception& e) {
    return error_json("invalid", e.what());
  }
}

extern "C" char* ksched_solve(void* session, int want_results) {
  Session* S = (Session*)session;
  if (!S || !S->handle) return S ? session_error(S) : error_json("invalid", "null session");
  ksolve_results res{};
  ksolve_status st = S->api.solve(S->handle, &res);
  return results_json(S, res, st, want_results);
}

// Solve() for n sessions with ONE launch of the pack kernel (ksolve_solve_batch): every probe of a consolidation sweep
// is its own wavefront. outs receives n result documents (each to be released with ksched_free). Returns 0 on success.
extern "C" int ksched_solve_batch(void** sessions, int n, int want_results, char** outs) {
  if (!sessions || n <= 0 || !outs) return 1;
  std::vector<ksolve_handle*> hs;
  for (int i = 0; i < n; ++i) {
    Session* S = (Session*)sessions[i];
    outs[i] = nullptr;
    if (!S || !S->handle) { for (int j = 0; j < n; ++j) outs[j] = S ? session_error(S) : error_json("invalid", "null session"); return 1; }
    hs.push_back(S->handle);
  }
  Session* S0 = (Session*)sessions[0];
  auto batch = (decltype(&ksolve_solve_batch))dlsym(S0->api.lib, "ksolve_solve_batch");
  if (!batch) { for (int j = 0; j < n; ++j) outs[j] = error_json("load", "solver library lacks ksolve_solve_batch"); return 1; }
  std::vector<ksolve_results> res((size_t)n);
  batch(hs.data(), (uint32_t)n, res.data());
  for (int i = 0; i < n; ++i) outs[i] = results_json((Session*)sessions[i], res[i], res[i].status, want_results);
  return 0;
}

// Convenience: open + solve (repeat times, last result returned with every run's timings) + close.
// ---- NodePool components: the host-side split of one provisioning batch for several devices -----------------------
// Two NodePools are in one component when some pod may land on either (its REQUIRED constraints on karpenter.sh/nodepool: a node
// selector and/or every required node-affinity term carrying In [...] on that key — pkg/utils/nodepool/nodepool.go:161-171 orders the
// pools, provisioner.go:293-297 hands all of them to one Scheduler), or when a topology group a pod of one owns (spread constraint,
// pod affinity / anti-affinity term, topology.go:461-533) selects a pod of the other: its domain counts move with every selected pod
// that is placed (topology.go:197-224). Returns {"components": [{"pools", "pods", "bin", "problem"}], "bins": [[component index]],
// "binPods": [pods per bin]} — components in NodePool order, dealt over `n_bins` devices by pod count, largest first (LPT) — or
// {"components": null, "reason"} when independence cannot be shown (existing nodes, cluster pods, reserved capacity, a pod that is not
// provably pinned, a selector this code cannot evaluate). Each component is a packing problem of its own, solved bit-exactly as such;
// the union is a packing of equal quality, NOT the reference's pod-for-pod answer for the whole batch.
static const char* kNodePoolKey = "karpenter.sh/nodepool";
struct SplitRefusal { std::string why; };
static std::vector<int> split_pinned_pools(const Value& p, const std::vector<std::string>& names) {
  auto idx = [&](const std::string& n) { for (size_t i = 0; i < names.size(); ++i) if (names[i] == n) return (int)i; return -1; };
  bool have = false;
  std::vector<char> allowed(names.size(), 0);
  const Value& sel = p.at("nodeSelector").at(kNodePoolKey);
  if (!sel.is_null()) { have = true; const int i = idx(sel.s()); if (i >= 0) allowed[(size_t)i] = 1; }
  const auto& terms = p.at("nodeAffinity").at("required").items();
  if (!terms.empty()) {
    std::vector<char> uni(names.size(), 0);
    bool pinned = true;
    for (auto& term : terms) {
      std::vector<char> in_all(names.size(), 1);
      bool any = false;
      for (auto& q : term.items()) if (q.at("key").s() == kNodePoolKey && q.at("operator").s() == "In") {
        any = true;
        std::vector<char> in(names.size(), 0);
        for (auto& v : q.at("values").items()) { const int i = idx(v.s()); if (i >= 0) in[(size_t)i] = 1; }
        for (size_t i = 0; i < names.size(); ++i) in_all[i] = in_all[i] && in[i];
      }
      if (!any) { pinned = false; break; }     // a term without the pin can reach any pool
      for (size_t i = 0; i < names.size(); ++i) uni[i] = uni[i] || in_all[i];
    }
    if (pinned) {
      if (!have) { allowed = uni; have = true; }
      else for (size_t i = 0; i < names.size(); ++i) allowed[i] = allowed[i] && uni[i];
    }
  }
  std::vector<int> out;
  if (have) for (size_t i = 0; i < names.size(); ++i) if (allowed[i]) out.push_back((int)i);
  return out;
}
struct SplitSelector { std::vector<std::string> namespaces; std::vector<std::pair<std::string, std::string>> match; };
static bool split_selectors(const Value& p, std::vector<SplitSelector>& out) {   // false: a term this code cannot evaluate
  const std::string ns = p.at("namespace").s("default");
  // (a group on any key but the hostname draws its domain universe from EVERY NodePool — buildDomainGroups, topology.go:104-142;
  // domainMinCount takes the minimum over all of them, topologygroup.go:300-322 — so its owner cannot be cut off from the other pools:
  // the whole batch may reject a pod a component would schedule. Such a term refuses the split. ADVICE r5.)
  for (auto& c : p.at("topologySpreadConstraints").items()) {
    const Value& sel = c.at("labelSelector");
    if (c.at("topologyKey").s() != "kubernetes.io/hostname") return false;
    if (!sel.at("matchExpressions").items().empty()) return false;
    SplitSelector s; s.namespaces = {ns};
    for (auto& kv : sel.at("matchLabels").members()) s.match.push_back({kv.first, kv.second.s()});
    out.push_back(s);
  }
  for (const char* field : {"podAffinity", "podAntiAffinity"}) {
    const Value& aff = p.at(field);
    std::vector<const Value*> terms;
    for (auto& t : aff.at("required").items()) terms.push_back(&t);
    for (auto& w : aff.at("preferred").items()) terms.push_back(&w.at("term"));
    for (const Value* t : terms) {
      const Value& sel = t->at("labelSelector");
      if (t->at("topologyKey").s() != "kubernetes.io/hostname") return false;
      if (!sel.at("matchExpressions").items().empty() || !t->at("namespaceSelector").is_null()) return false;
      SplitSelector s;
      for (auto& n : t->at("namespaces").items()) s.namespaces.push_back(n.s());
      if (s.namespaces.empty()) s.namespaces = {ns};
      for (auto& kv : sel.at("matchLabels").members()) s.match.push_back({kv.first, kv.second.s()});
      out.push_back(s);
    }
  }
  return true;
}
extern "C" char* ksched_split_components(const char* problem_json, int n_bins) {
  try {
    const Value prob = kj::Parser(problem_json).parse();
    auto refuse = [](const std::string& why) { Value o = Value::object(); o.set("components", Value()); o.set("reason", Value::string(why)); return dup_json(o); };
    if (!prob.at("stateNodes").items().empty()) return refuse("existing nodes are bins every NodePool's pods share");
    if (!prob.at("clusterPods").items().empty()) return refuse("cluster pods: shared topology counts");
    if (prob.at("options").at("reservedCapacity").boolean_or(false)) return refuse("reservations are shared between NodePools");
    std::vector<std::string> names;
    for (auto& np : prob.at("nodePools").items()) names.push_back(np.at("name").s());
    const int P = (int)names.size();
    std::vector<int> parent((size_t)P);
    for (int i = 0; i < P; ++i) parent[(size_t)i] = i;
    std::function<int(int)> find = [&](int x) { while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; } return x; };
    struct Item { bool group; const Value* v; const Value* tmpl; int first; int64_t pods; };
    std::vector<Item> items;
    for (int kind = 0; kind < 2; ++kind) for (auto& it : prob.at(kind ? "podGroups" : "pods").items()) {
      const Value& t = kind ? it.at("template") : it;
      const std::vector<int> allowed = split_pinned_pools(t, names);
      if (allowed.empty()) return refuse("a pod is not provably pinned to NodePools by its required constraints on karpenter.sh/nodepool");
      for (int o : allowed) parent[(size_t)find(o)] = find(allowed[0]);
      items.push_back({kind != 0, &it, &t, allowed[0], kind ? it.at("count").i(0) : 1});
    }
    // topology groups tie their owner to every pod they select: pods by (namespace, labels) signature
    struct Sig { std::string ns; const Value* labels; std::vector<int> firsts; };
    std::vector<Sig> sigs;
    {
      std::map<std::string, size_t> seen;
      for (auto& it : items) {
        std::string key = it.tmpl->at("namespace").s("default") + "\x01";
        std::vector<std::pair<std::string, std::string>> ls;
        for (auto& kv : it.tmpl->at("labels").members()) ls.push_back({kv.first, kv.second.s()});
        std::sort(ls.begin(), ls.end());
        for (auto& kv : ls) key += kv.first + "\x02" + kv.second + "\x03";
        auto f = seen.find(key);
        if (f == seen.end()) { seen[key] = sigs.size(); sigs.push_back({it.tmpl->at("namespace").s("default"), &it.tmpl->at("labels"), {it.first}}); }
        else sigs[f->second].firsts.push_back(it.first);
      }
    }
    for (auto& it : items) {
      std::vector<SplitSelector> sels;
      if (!split_selectors(*it.tmpl, sels)) return refuse("a topology group on a key other than kubernetes.io/hostname (its domains come from every NodePool), or a topology selector with matchExpressions / a namespaceSelector");
      for (auto& sl : sels) for (auto& sg : sigs) {
        if (std::find(sl.namespaces.begin(), sl.namespaces.end(), sg.ns) == sl.namespaces.end()) continue;
        bool all = true;
        for (auto& kv : sl.match) { const Value& v = sg.labels->at(kv.first); if (v.is_null() || v.s() != kv.second) { all = false; break; } }
        if (all) for (int f : sg.firsts) parent[(size_t)find(f)] = find(it.first);
      }
    }
    // components in NodePool order
    std::vector<int> comp_of((size_t)P, -1);
    std::vector<std::vector<int>> comp_pools;
    for (int i = 0; i < P; ++i) { const int r = find(i); if (comp_of[(size_t)r] < 0) { comp_of[(size_t)r] = (int)comp_pools.size(); comp_pools.push_back({}); } comp_pools[(size_t)comp_of[(size_t)r]].push_back(i); }
    const size_t C = comp_pools.size();
    std::vector<Value> pods, groups;   // (one array each: a copied Value shares its array)
    for (size_t c = 0; c < C; ++c) { pods.push_back(Value::array()); groups.push_back(Value::array()); }
    std::vector<int64_t> count(C, 0);
    for (auto& it : items) { const size_t c = (size_t)comp_of[(size_t)find(it.first)]; (it.group ? groups[c] : pods[c]).push(*it.v); count[c] += it.pods; }
    std::vector<size_t> keep;
    for (size_t c = 0; c < C; ++c) if (count[c] > 0) keep.push_back(c);
    // LPT: the component with the most pods first, each to the bin with the fewest pods so far (lowest index on ties)
    const int B = n_bins < 1 ? 1 : n_bins;
    std::vector<int64_t> load((size_t)B, 0);
    std::vector<int> bin_of(C, 0);
    std::vector<size_t> by_size = keep;
    std::stable_sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return count[a] > count[b]; });
    for (size_t c : by_size) { int best = 0; for (int b = 1; b < B; ++b) if (load[(size_t)b] < load[(size_t)best]) best = b; bin_of[c] = best; load[(size_t)best] += count[c]; }
    Value out = Value::object(), comps = Value::array(), bins = Value::array(), bin_pods = Value::array();
    std::vector<Value> bin_lists;
    for (int b = 0; b < B; ++b) bin_lists.push_back(Value::array());
    for (size_t k = 0; k < keep.size(); ++k) {
      const size_t c = keep[k];
      Value sub = Value::object();
      for (auto& kv : prob.members()) {
        if (kv.first == "nodePools") { Value nps = Value::array(); for (int i : comp_pools[c]) nps.push(prob.at("nodePools").items()[(size_t)i]); sub.add_new("nodePools", nps); }
        else if (kv.first == "pods") sub.add_new("pods", pods[c]);
        else if (kv.first == "podGroups") sub.add_new("podGroups", groups[c]);
        else sub.add_new(kv.first, kv.second);
      }
      Value e = Value::object(), pn = Value::array();
      for (int i : comp_pools[c]) pn.push(Value::string(names[(size_t)i]));
      e.set("pools", pn); e.set("pods", Value::integer(count[c])); e.set("bin", Value::integer(bin_of[c])); e.set("problem", sub);
      comps.push(e);
      bin_lists[(size_t)bin_of[c]].push(Value::integer((int64_t)k));
    }
    for (int b = 0; b < B; ++b) { bins.push(bin_lists[(size_t)b]); bin_pods.push(Value::integer(load[(size_t)b])); }
    out.set("components", comps); out.set("bins", bins); out.set("binPods", bin_pods);
    return dup_json(out);
  } catch (const std::exception& e) {
    return error_json("invalid", e.what());
  }
}

extern "C" char* ksched_solve_json(const char* problem_json, const char* solver_lib, int repeat, int want_results) {
  void* s = ksched_open(problem_json, solver_lib);
  if (ksched_error(s)) { char* e = session_error((Session*)s); ksched_close(s); return e; }
  char* out = nullptr;
  std::vector<std::string> timings;
  for (int i = 0; i < std::max(1, repeat); ++i) {
    if (out) free(out);
    out = ksched_solve(s, want_results);
    Value v = kj::Parser(out).parse();
    if (v.has("error")) break;
    std::string t;
    kj::write(v.at("timings").items()[0], t);
    timings.push_back(t);
  }
  ksched_close(s);
  Value v = kj::Parser(out).parse();
  if (!v.has("error")) {
    Value all = Value::array();
    for (auto& t : timings) all.push(kj::Parser(t.c_str()).parse());
    v.set("timings", all);
    free(out);
    out = dup_json(v);
  }
  return out;
}
