// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
//
// String interning for the CPU restatement (round 4). The oracle used to keep every label key, label value, pod UID and
// resource name as a std::string inside std::map / std::set; one NodeClaim.CanAdd then cost ~30 red-black-tree walks over
// strings and the configs[2] shape could not be pinned above 300k pods (5.4 h). Strings are now interned ONCE, at parse time
// (and when a hostname placeholder is minted), into dense int32 symbols; sets of symbols are small sorted arrays. This is a
// change of container, not of algorithm: every function still walks the reference's control flow and cites it. Nothing here
// is shared with the product (karpenter_amd/ encodes values as bit positions of per-key dictionaries; the oracle has no
// dictionaries, no bitmasks and no notion of a pod class) — wherever the reference's result depends on an ORDER of strings
// (canonical topology ties, Requirement.Any, output), the strings themselves are compared (sym_lex_less).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace oracle {

typedef int32_t Sym;
constexpr Sym kNoSym = -1;

// Append-only, thread-safe (oracle_sweep_json simulates probes on several threads and each mints hostname placeholders).
class Interner {
  static constexpr int kBlockBits = 12;
  static constexpr int kBlock = 1 << kBlockBits;
  static constexpr int kMaxBlocks = 1 << 15;   // 134M strings
  std::atomic<std::string*> blocks_[kMaxBlocks];
  std::unordered_map<std::string, Sym> index_;
  mutable std::shared_mutex mu_;
  int32_t n_ = 0;

 public:
  Interner() { for (auto& b : blocks_) b.store(nullptr, std::memory_order_relaxed); }
  Sym id(const std::string& s) {
    {
      std::shared_lock<std::shared_mutex> l(mu_);
      auto it = index_.find(s);
      if (it != index_.end()) return it->second;
    }
    std::unique_lock<std::shared_mutex> l(mu_);
    auto it = index_.find(s);
    if (it != index_.end()) return it->second;
    const int32_t i = n_;
    const int b = i >> kBlockBits;
    if (b >= kMaxBlocks) abort();
    std::string* blk = blocks_[b].load(std::memory_order_relaxed);
    if (!blk) { blk = new std::string[kBlock]; blocks_[b].store(blk, std::memory_order_release); }
    blk[i & (kBlock - 1)] = s;
    index_.emplace(s, i);
    n_ = i + 1;
    return i;
  }
  const std::string& str(Sym s) const { return blocks_[s >> kBlockBits].load(std::memory_order_acquire)[s & (kBlock - 1)]; }
};
inline Interner& interner() { static Interner* in = new Interner(); return *in; }
inline Sym sym(const std::string& s) { return interner().id(s); }
inline const std::string& str(Sym s) { return interner().str(s); }
// lexicographic order of the strings behind two symbols (std::map<std::string> / std::set<std::string> iteration order)
inline bool sym_lex_less(Sym a, Sym b) { return a != b && str(a) < str(b); }
struct SymLexLess { bool operator()(Sym a, Sym b) const { return sym_lex_less(a, b); } };

// A vector with inline storage for trivially copyable elements (symbol sets and resource lists are a handful of entries;
// NodeClaim.CanAdd copies several of them per call).
template <class T, unsigned N>
class SmallVec {
  T* p_;
  uint32_t n_, cap_;
  T inl_[N];
  void grow(uint32_t want) {
    uint32_t c = cap_ * 2 > want ? cap_ * 2 : want;
    T* np = (T*)malloc(sizeof(T) * c);
    memcpy((void*)np, (const void*)p_, sizeof(T) * n_);
    if (p_ != inl_) free(p_);
    p_ = np;
    cap_ = c;
  }

 public:
  typedef T* iterator;
  typedef const T* const_iterator;
  SmallVec() : p_(inl_), n_(0), cap_(N) {}
  SmallVec(const SmallVec& o) : p_(inl_), n_(0), cap_(N) { assign(o.p_, o.n_); }
  SmallVec(SmallVec&& o) noexcept : p_(inl_), n_(o.n_), cap_(N) {
    if (o.p_ != o.inl_) { p_ = o.p_; cap_ = o.cap_; o.p_ = o.inl_; o.cap_ = N; }
    else memcpy((void*)inl_, (const void*)o.inl_, sizeof(T) * n_);
    o.n_ = 0;
  }
  SmallVec& operator=(const SmallVec& o) { if (this != &o) assign(o.p_, o.n_); return *this; }
  SmallVec& operator=(SmallVec&& o) noexcept {
    if (this == &o) return *this;
    if (p_ != inl_) free(p_);
    p_ = inl_; cap_ = N; n_ = o.n_;
    if (o.p_ != o.inl_) { p_ = o.p_; cap_ = o.cap_; o.p_ = o.inl_; o.cap_ = N; }
    else memcpy((void*)inl_, (const void*)o.inl_, sizeof(T) * n_);
    o.n_ = 0;
    return *this;
  }
  ~SmallVec() { if (p_ != inl_) free(p_); }
  void assign(const T* src, uint32_t n) {
    if (n > cap_) { if (p_ != inl_) free(p_); p_ = (T*)malloc(sizeof(T) * n); cap_ = n; }
    memcpy((void*)p_, (const void*)src, sizeof(T) * n);
    n_ = n;
  }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  void clear() { n_ = 0; }
  T* begin() { return p_; }
  T* end() { return p_ + n_; }
  const T* begin() const { return p_; }
  const T* end() const { return p_ + n_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }
  T& back() { return p_[n_ - 1]; }
  void push_back(const T& v) { if (n_ == cap_) grow(n_ + 1); p_[n_++] = v; }
  void insert_at(size_t pos, const T& v) {
    if (n_ == cap_) grow(n_ + 1);
    memmove((void*)(p_ + pos + 1), (const void*)(p_ + pos), sizeof(T) * (n_ - pos));
    p_[pos] = v;
    n_++;
  }
  void erase_at(size_t pos) { memmove((void*)(p_ + pos), (const void*)(p_ + pos + 1), sizeof(T) * (n_ - pos - 1)); n_--; }
  void reserve(uint32_t c) { if (c > cap_) grow(c); }
};

inline std::vector<Sym> lex_sort(std::vector<Sym> v) {
  std::sort(v.begin(), v.end(), [](Sym a, Sym b) { return str(a) < str(b); });
  return v;
}

// A set of symbols (sets.Set[string] in the reference): unique, sorted by symbol id. Iteration order is NOT lexicographic;
// code whose result depends on string order says so and sorts (lex_sorted / lex_min).
class SymSet {
  SmallVec<Sym, 6> v_;
  size_t lower(Sym s) const {
    size_t lo = 0, hi = v_.size();
    if (hi <= 8) { while (lo < hi && v_[lo] < s) ++lo; return lo; }
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (v_[mid] < s) lo = mid + 1; else hi = mid; }
    return lo;
  }

 public:
  SymSet() {}
  SymSet(std::initializer_list<Sym> l) { for (Sym s : l) insert(s); }
  size_t size() const { return v_.size(); }
  bool empty() const { return v_.empty(); }
  const Sym* begin() const { return v_.begin(); }
  const Sym* end() const { return v_.end(); }
  bool count(Sym s) const { size_t i = lower(s); return i < v_.size() && v_[i] == s; }
  void insert(Sym s) { size_t i = lower(s); if (i == v_.size() || v_[i] != s) v_.insert_at(i, s); }
  template <class It> void insert(It a, It b) { for (; a != b; ++a) insert(*a); }
  void erase(Sym s) { size_t i = lower(s); if (i < v_.size() && v_[i] == s) v_.erase_at(i); }
  void clear() { v_.clear(); }
  void append_sorted(Sym s) { v_.push_back(s); }   // caller guarantees s > every element
  bool operator==(const SymSet& o) const { return v_.size() == o.v_.size() && memcmp(v_.begin(), o.v_.begin(), sizeof(Sym) * v_.size()) == 0; }
  bool operator!=(const SymSet& o) const { return !(*this == o); }
  // lexicographically smallest string of the set (first element of a std::set<std::string>)
  Sym lex_min() const {
    Sym best = kNoSym;
    for (Sym s : v_) if (best == kNoSym || sym_lex_less(s, best)) best = s;
    return best;
  }
  std::vector<Sym> lex_sorted() const;
  std::vector<std::string> strings() const { std::vector<std::string> out; for (Sym s : lex_sorted()) out.push_back(str(s)); return out; }
};
inline std::vector<Sym> SymSet::lex_sorted() const { return lex_sort(std::vector<Sym>(begin(), end())); }

// map[string]string as a small sorted array of (key, value) symbols (pod / node labels, selectors' matchLabels)
struct SymPair { Sym first, second; };
class SymMap {
  SmallVec<SymPair, 6> v_;

 public:
  size_t size() const { return v_.size(); }
  bool empty() const { return v_.empty(); }
  const SymPair* begin() const { return v_.begin(); }
  const SymPair* end() const { return v_.end(); }
  const SymPair* find(Sym k) const { for (auto& e : v_) if (e.first == k) return &e; return nullptr; }
  bool count(Sym k) const { return find(k) != nullptr; }
  void set(Sym k, Sym val) {
    size_t i = 0;
    while (i < v_.size() && v_[i].first < k) ++i;
    if (i < v_.size() && v_[i].first == k) v_[i].second = val;
    else v_.insert_at(i, SymPair{k, val});
  }
  bool operator==(const SymMap& o) const { return v_.size() == o.v_.size() && memcmp(v_.begin(), o.v_.begin(), sizeof(SymPair) * v_.size()) == 0; }
};

}  // namespace oracle
