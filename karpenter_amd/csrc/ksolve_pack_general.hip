// ksolve_pack_general.hip — the general engine (engine.h) on ONE problem per launch: full, lite and BIG instantiations.
#include "pack_kernels.h"

// One wavefront (64 threads) per problem; the working requirement set and candidate lists live in LDS.
__global__ void __launch_bounds__(64) ksolve_pack(ks::ProblemView pv, ks::Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  if (pv.lds.topo_bytes) tables.topo = lds + pv.lds.off_topo;   // (one problem per launch: the topology groups' descriptors and small state in LDS, engine.h topo_to_lds)
  ks::Engine<ks::Wave, true> eng(pv, ws, tables);
  eng.solve();
}
// The same engine compiled without topology / existing nodes / daemon overhead / minValues / reservations, for problems
// that use none of them (ProblemView::lite): less code and far less live state around the hot loop.
__global__ void __launch_bounds__(64) ksolve_pack_lite(ks::ProblemView pv, ks::Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  ks::Engine<ks::Wave, false> eng(pv, ws, tables);
  eng.solve();
}

// The full engine with the claim order in HBM, for problems with more in-flight claims than a CU's LDS can order
// (every anti-affinity / hostname-spread pod is its own NodeClaim): ProblemView::big.
__global__ void __launch_bounds__(64) ksolve_pack_big(ks::ProblemView pv, ks::Workspace ws) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, pv.lds);
  if (pv.lds.topo_bytes) tables.topo = lds + pv.lds.off_topo;   // (one problem per launch: the topology groups' descriptors and small state in LDS, engine.h topo_to_lds)
  ks::Engine<ks::Wave, true, true> eng(pv, ws, tables);
  eng.solve();
}
