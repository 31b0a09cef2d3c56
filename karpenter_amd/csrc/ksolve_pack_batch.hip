// ksolve_pack_batch.hip — the general engine on many problems per launch: batches and the one-wavefront consolidation sweep.
#include "pack_kernels.h"

// Batched form: block b solves problem b (its view and workspace are read from HBM instead of the kernel arguments).
__global__ void __launch_bounds__(64) ksolve_pack_batch(ks::BatchItem* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::BatchItem& it = items[blockIdx.x];
  ks::LdsTables tables;
  tables.bind(lds, it.pv.lds);
  ks::Engine<ks::Wave, true> eng(it.pv, it.ws, tables);
  eng.solve();
}
__global__ void __launch_bounds__(64) ksolve_pack_batch_lite(ks::BatchItem* items) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::BatchItem& it = items[blockIdx.x];
  ks::LdsTables tables;
  tables.bind(lds, it.pv.lds);
  ks::Engine<ks::Wave, false> eng(it.pv, it.ws, tables);
  eng.solve();
}
// A consolidation sweep over a resident cluster: block b runs the general engine on probes b, b + gridDim.x, ... — one view of
// the cluster for all of them (HBM), one workspace per probe (its claims and its node overlay), one LDS plan for the launch.
__global__ void __launch_bounds__(64) ksolve_pack_sweep(const ks::ProblemView* pv, ks::Workspace* items, int n, ks::LdsPlan plan) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  ks::LdsTables tables;
  tables.bind(lds, plan);
  for (int p = (int)blockIdx.x; p < n; p += (int)gridDim.x) {
    ks::Engine<ks::Wave, true> eng(*pv, items[p], tables);
    eng.solve();
  }
}
