// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called by the product path.
// Minimal JSON value + parser + writer used by the CPU oracle to read problem files and emit results.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace oj {

struct Value;
using Array = std::vector<Value>;
using Object = std::vector<std::pair<std::string, Value>>;  // insertion ordered

struct Value {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  bool b = false;
  double num = 0;
  bool is_int = false;
  int64_t inum = 0;
  std::string str;
  std::shared_ptr<Array> arr;
  std::shared_ptr<Object> obj;

  Value() {}
  static Value boolean(bool v) { Value x; x.kind = Bool; x.b = v; return x; }
  static Value number(double v) { Value x; x.kind = Num; x.num = v; return x; }
  static Value integer(int64_t v) { Value x; x.kind = Num; x.num = (double)v; x.is_int = true; x.inum = v; return x; }
  static Value string(const std::string& s) { Value x; x.kind = Str; x.str = s; return x; }
  static Value array() { Value x; x.kind = Arr; x.arr = std::make_shared<Array>(); return x; }
  static Value object() { Value x; x.kind = Obj; x.obj = std::make_shared<Object>(); return x; }

  bool is_null() const { return kind == Null; }
  bool has(const std::string& k) const {
    if (kind != Obj) return false;
    for (auto& kv : *obj) if (kv.first == k) return true;
    return false;
  }
  const Value& at(const std::string& k) const {
    static Value nullv;
    if (kind != Obj) return nullv;
    for (auto& kv : *obj) if (kv.first == k) return kv.second;
    return nullv;
  }
  Value& set(const std::string& k, const Value& v) {
    for (auto& kv : *obj) if (kv.first == k) { kv.second = v; return kv.second; }
    obj->push_back({k, v});
    return obj->back().second;
  }
  void push(const Value& v) { arr->push_back(v); }
  const Array& items() const { static Array e; return kind == Arr ? *arr : e; }
  const Object& members() const { static Object e; return kind == Obj ? *obj : e; }
  std::string s(const std::string& dflt = "") const { return kind == Str ? str : dflt; }
  int64_t i(int64_t dflt = 0) const { return kind == Num ? (is_int ? inum : (int64_t)num) : dflt; }
  double d(double dflt = 0) const { return kind == Num ? num : dflt; }
  bool boolean_or(bool dflt) const { return kind == Bool ? b : dflt; }
};

class Parser {
 public:
  explicit Parser(const char* p) : p_(p) {}
  Value parse() { Value v = value(); ws(); if (*p_) fail("trailing"); return v; }

 private:
  const char* p_;
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m); }
  void ws() { while (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r') ++p_; }
  Value value() {
    ws();
    switch (*p_) {
      case '{': return object();
      case '[': return array();
      case '"': return Value::string(str());
      case 't': expect("true"); return Value::boolean(true);
      case 'f': expect("false"); return Value::boolean(false);
      case 'n': expect("null"); return Value();
      default: return number();
    }
  }
  void expect(const char* lit) { for (; *lit; ++lit, ++p_) if (*p_ != *lit) fail("literal"); }
  static bool digit(char c) { return c >= '0' && c <= '9'; }
  // RFC 8259 number: -? (0 | [1-9][0-9]*) (\.[0-9]+)? ([eE][+-]?[0-9]+)? — an integer that does not fit 64 bits becomes a double
  Value number() {
    const char* s = p_;
    bool isint = true;
    if (*p_ == '-') ++p_;
    if (*p_ == '0') ++p_;
    else if (*p_ >= '1' && *p_ <= '9') { while (digit(*p_)) ++p_; }
    else fail("number");
    if (*p_ == '.') { isint = false; ++p_; if (!digit(*p_)) fail("number"); while (digit(*p_)) ++p_; }
    if (*p_ == 'e' || *p_ == 'E') { isint = false; ++p_; if (*p_ == '+' || *p_ == '-') ++p_; if (!digit(*p_)) fail("number"); while (digit(*p_)) ++p_; }
    std::string t(s, p_);
    if (isint) {
      errno = 0;
      const long long iv = strtoll(t.c_str(), nullptr, 10);
      if (errno != ERANGE) { Value v = Value::integer(iv); v.num = strtod(t.c_str(), nullptr); return v; }
    }
    return Value::number(strtod(t.c_str(), nullptr));
  }
  static int hexv(char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1; }
  unsigned hex4() {   // p_ at the 'u' of an escape; leaves it at the last of the four digits
    unsigned cp = 0;
    for (int k = 1; k <= 4; ++k) { const int h = hexv(p_[k]); if (h < 0) fail("\\u escape"); cp = cp * 16 + (unsigned)h; }
    p_ += 4;
    return cp;
  }
  std::string str() {
    ++p_;
    std::string out;
    while (*p_ && *p_ != '"') {
      if (*p_ == '\\') {
        ++p_;
        switch (*p_) {
          case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
          case 'b': out += '\b'; break; case 'f': out += '\f'; break;
          case '"': case '\\': case '/': out += *p_; break;
          case 'u': {
            unsigned cp = hex4();
            if (cp >= 0xD800 && cp <= 0xDBFF && p_[1] == '\\' && p_[2] == 'u') {   // a surrogate pair: ONE code point beyond the BMP (found by tests/test_json_parsers.py: the pair used to come out as two three-byte sequences, which is not UTF-8)
              const char* save = p_;
              p_ += 2;
              const unsigned lo = hex4();
              if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
              else p_ = save;
            }
            if (cp >= 0xD800 && cp <= 0xDFFF) fail("lone surrogate");   // (an unpaired \uD800-\uDFFF has no UTF-8 form: refused, like Python's json in strict UTF-8 output — ADVICE r5)
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: fail("escape");
        }
        ++p_;
      } else out += *p_++;
    }
    if (*p_ != '"') fail("string");
    ++p_;
    return out;
  }
  Value array() {
    Value v = Value::array();
    ++p_; ws();
    if (*p_ == ']') { ++p_; return v; }
    for (;;) {
      v.push(value()); ws();
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == ']') { ++p_; return v; }
      fail("array");
    }
  }
  Value object() {
    Value v = Value::object();
    ++p_; ws();
    if (*p_ == '}') { ++p_; return v; }
    for (;;) {
      ws(); if (*p_ != '"') fail("key");
      std::string k = str(); ws();
      if (*p_ != ':') fail("colon");
      ++p_;
      v.obj->push_back({k, value()}); ws();
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == '}') { ++p_; return v; }
      fail("object");
    }
  }
};

inline void write(const Value& v, std::string& out) {
  switch (v.kind) {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v.b ? "true" : "false"; break;
    case Value::Num: {
      char buf[40];
      if (v.is_int) snprintf(buf, sizeof buf, "%lld", (long long)v.inum);
      else snprintf(buf, sizeof buf, "%.17g", v.num);
      out += buf; break;
    }
    case Value::Str: {
      out += '"';
      for (unsigned char c : v.str) {
        if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
        else if (c == '\n') out += "\\n";
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
        else out += (char)c;
      }
      out += '"'; break;
    }
    case Value::Arr: {
      out += '[';
      bool first = true;
      for (auto& x : *v.arr) { if (!first) out += ','; first = false; write(x, out); }
      out += ']'; break;
    }
    case Value::Obj: {
      out += '{';
      bool first = true;
      for (auto& kv : *v.obj) { if (!first) out += ','; first = false; write(Value::string(kv.first), out); out += ':'; write(kv.second, out); }
      out += '}'; break;
    }
  }
}

}  // namespace oj
