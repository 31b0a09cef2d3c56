// topo_types.h — what the host side (ksolve_impl.h: buffers, LDS plan, kernel arguments) shares with the spread engine
// (topo_engine.h, compiled into ksolve_pack_topo.hip and the test emulation only).
#pragma once
#include "fast_engine.h"
#include "run_order.h"

namespace ks {

constexpr int kTopoMaxGroups = 128;   // topology groups of a problem this engine takes
constexpr int kTopoMaxHost = 16;      // of them on kubernetes.io/hostname (regular + inverse): one 4-bit field each
constexpr int kTopoMaxZg = 64;        // ... and on dictionary keys
constexpr int kTopoMaxDom = 16;       // domains of such a key
constexpr int kTopoTrack = 4;         // anti-affinity counters with a list of the claims that hold no member
constexpr int kTopoFreeCap = 1024;    // entries of such a list (more: the list is dropped, its classes scan the order)
constexpr uint64_t kTopoGuard = 0x8888888888888888ull, kTopoOnes = 0x1111111111111111ull;

struct TopoRec { uint64_t vmask; int32_t req[4]; uint64_t hcnt; };   // 32 B: an in-flight claim (requirement set, requests, hostname-group counters)
// a pod class's topology: limits on the hostname counters it is tested against (field = 8 | limit; 8 | 7 where it has none),
// the counters and dictionary-key groups a pod of the class is counted by, the dictionary-key group it owns
struct TopoClass { uint64_t hlim, hinc, zsel; int16_t zg[2]; uint32_t zself; uint32_t excl; uint32_t pad; };   // 48 B; zg: up to two dictionary-key groups it is tested against (-1: none), zself bit i: it is selected by zg[i] itself
struct TopoZg {   // a group on a dictionary key (LDS): sixteen counters and name ranks, then one 16-byte header
  int32_t cnt[kTopoMaxDom];
  uint16_t rank[kTopoMaxDom];
  uint32_t dom;        // registered domains (TopologyGroup.domains)
  int32_t nonzero;     // domains with a positive count
  int16_t skew, min_domains;   // maxSkew; minDomains (-1: nil)
  uint8_t type, var, off, width;   // 0 spread / 1 affinity / 2 anti-affinity (also the inverse groups); index of its key among the variable keys (FastMisc::vkey); the key's field inside vmask
};
struct TopoZgHead { uint32_t dom; int32_t nonzero; int16_t skew, min_domains; uint8_t type, var, off, width; };   // TopoZg from `dom` on
static_assert(sizeof(TopoZg) == 112 && sizeof(TopoZgHead) == 16, "TopoZg layout");
struct TopoState {   // LDS
  TopoZg zg[kTopoMaxZg];
  uint32_t freel[kTopoTrack][kTopoFreeCap];
  int32_t n_free[kTopoTrack];
  uint32_t track_field[kTopoTrack];   // hostname counter of list t; 0xFF: none / dropped
  int16_t gmap[kTopoMaxGroups];       // group -> hostname counter (0..15) | 0x100 + dictionary-key group | -1
  // what the loop needs once per block of 64 pods or per new claim, kept out of its scalar registers
  const uint32_t* q_class; uint32_t* q_claim; uint32_t* q_cnt; const TopoClass* tcls; const FastSlot* fcls; uint32_t* hostseq;
  int32_t last[4];                    // the move of the last step: kind (1: a claim gained a pod, 2: a new claim, 0: pending in the order / none), claim, position it left
};
struct TopoPlan { int total_bytes, off_run, off_state; };   // (the cursor engine's tables sit where FastWork::plan says)
struct TopoWork { TopoClass* cls; TopoRec* rec; TopoPlan plan; int enabled; };
struct TopoArgs { ProblemView pv; Workspace ws; FastWork fw; TopoWork tw; };

}  // namespace ks
