// tiny -finstrument-functions profiler: inclusive cycles + calls per function address
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include <dlfcn.h>
#include <cxxabi.h>
#include <x86intrin.h>
#include <pthread.h>
static pthread_t g_owner; static volatile bool g_have_owner = false;
struct Ent { uint64_t calls = 0, incl = 0, self = 0; };
static std::unordered_map<void*, Ent>* g_map;
static std::vector<std::pair<void*, uint64_t>>* g_stack;   // fn, t_enter
static std::vector<uint64_t>* g_child;
static volatile bool g_on = false;
extern "C" {
__attribute__((no_instrument_function)) void prof_start() { if (!g_map) { g_map = new std::unordered_map<void*, Ent>(); g_stack = new std::vector<std::pair<void*, uint64_t>>(); g_child = new std::vector<uint64_t>(); } g_map->clear(); g_stack->clear(); g_child->clear(); g_have_owner = false; g_on = true; }
__attribute__((no_instrument_function)) void prof_stop(const char* path) {
  g_on = false;
  std::vector<std::pair<void*, Ent>> v(g_map->begin(), g_map->end());
  std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.second.self > b.second.self; });
  FILE* f = fopen(path, "w");
  for (auto& e : v) {
    Dl_info di; const char* nm = "?";
    char* dem = nullptr;
    if (dladdr(e.first, &di) && di.dli_sname) { int st = 0; dem = abi::__cxa_demangle(di.dli_sname, nullptr, nullptr, &st); nm = dem ? dem : di.dli_sname; }
    fprintf(f, "%12llu calls %14llu self %14llu incl  %s\n", (unsigned long long)e.second.calls, (unsigned long long)e.second.self, (unsigned long long)e.second.incl, nm);
    free(dem);
  }
  fclose(f);
}
__attribute__((no_instrument_function)) void __cyg_profile_func_enter(void* fn, void*) {
  if (!g_on) return;
  if (!g_have_owner) { g_owner = pthread_self(); g_have_owner = true; }
  if (!pthread_equal(g_owner, pthread_self())) return;
  g_stack->push_back({fn, __rdtsc()}); g_child->push_back(0);
}
__attribute__((no_instrument_function)) void __cyg_profile_func_exit(void* fn, void*) {
  if (!g_on || !g_have_owner || !pthread_equal(g_owner, pthread_self()) || g_stack->empty()) return;
  const uint64_t t = __rdtsc();
  auto top = g_stack->back(); g_stack->pop_back();
  const uint64_t ch = g_child->back(); g_child->pop_back();
  const uint64_t d = t - top.second;
  Ent& e = (*g_map)[top.first]; e.calls++; e.incl += d; e.self += d - std::min(d, ch);
  if (!g_child->empty()) g_child->back() += d;
}
}
