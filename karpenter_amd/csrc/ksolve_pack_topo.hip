// ksolve_pack_topo.hip — the spread engine (topo_engine.h).
#include "pack_kernels.h"

#include "topo_engine.h"

// The spread engine (topo_engine.h): the cursor engine's shape plus topology spread / pod affinity on dictionary keys and spread /
// anti-affinity on the hostname — BASELINE configs[2]. One wavefront; claims in HBM (32 B each), the order's rings in HBM, the
// caches, the ring tables and the topology counters in LDS.
__global__ void __launch_bounds__(64) ksolve_pack_topo(const ks::TopoArgs* a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::TopoEngine<ks::Wave> eng(&a->pv, &a->ws, &a->fw, &a->tw, lds);
  eng.solve();
}
