// ksp.h — device-side view of a flattened scheduling problem and of the solver workspace.
// Plain pointers into device memory (or host memory in the test-only emulation build). Layouts are SoA where a
// kernel streams over many entities (pods) and AoS-by-entity where one wave touches one entity at a time (claims).
#pragma once
#include "reqalg.h"

namespace ks {

struct ReqTable {  // n requirement sets, SoA (ksolve_reqsets)
  const uint64_t* mask;
  const uint32_t *defined, *complement, *has_gte, *has_lte;
  const int64_t *gte, *lte;
  const int32_t* minv;
  KS_FN ReqRef at(const Dict& d, uint32_t e) const {
    ReqRef r;
    r.mask = mask + (size_t)e * d.req_words;
    r.defined = defined[e]; r.complement = complement[e];
    r.has_gte = has_gte ? has_gte[e] : 0; r.has_lte = has_lte ? has_lte[e] : 0;
    r.gte = gte ? gte + (size_t)e * d.n_keys : nullptr;
    r.lte = lte ? lte + (size_t)e * d.n_keys : nullptr;
    r.minv = minv ? minv + (size_t)e * d.n_keys : nullptr;
    return r;
  }
};

struct MutReqTable {  // claims: same layout, writable
  uint64_t* mask;
  uint32_t *defined, *complement, *has_gte, *has_lte;
  int64_t *gte, *lte;
  int32_t* minv;
  KS_FN ReqRef at(const Dict& d, uint32_t e) const {
    ReqRef r;
    r.mask = mask + (size_t)e * d.req_words;
    r.defined = defined[e]; r.complement = complement[e]; r.has_gte = has_gte[e]; r.has_lte = has_lte[e];
    r.gte = gte + (size_t)e * d.n_keys; r.lte = lte + (size_t)e * d.n_keys; r.minv = minv + (size_t)e * d.n_keys;
    return r;
  }
};

// Record layouts (u64 words). A claim / class / template is moved between HBM and LDS as ONE contiguous record with a
// single coalesced wave load (lane i moves word i), instead of many dependent scalar loads.
//   hot  claim record : masks[rw] | its[iw] | total[nr] | head[nr] | f0 | f1 | meta | meta2
//                       f0 = defined | complement<<32 ; f1 = has_gte | has_lte<<32 ; meta = template | npods<<32 ;
//                       meta2 = hostname seq | flags<<32 (bit0 min-values-relaxed, bit1 has_minv)
//   cold claim record : gte[nk] | lte[nk] | minv[nk] as packed i32               — only touched when bounds/minValues exist
//   hot  class record : masks[rw] | requests[nr] | f0 | f1 | tolerates | meta(has_minv bit0)
//   cold class record : gte[nk] | lte[nk] | minv[nk]
struct RecLayout {
  int rw, iw, nr, nk;
  KS_FN int c_mask() const { return 0; }
  KS_FN int c_its() const { return rw; }
  KS_FN int c_total() const { return rw + iw; }
  KS_FN int c_head() const { return rw + iw + nr; }
  KS_FN int c_f0() const { return rw + iw + 2 * nr; }
  KS_FN int c_f1() const { return c_f0() + 1; }
  KS_FN int c_meta() const { return c_f0() + 2; }
  KS_FN int c_meta2() const { return c_f0() + 3; }
  KS_FN int c_hot_words() const { return c_f0() + 4; }
  KS_FN int cold_words() const { return 2 * nk + (nk + 1) / 2; }
  KS_FN int k_mask() const { return 0; }
  KS_FN int k_req() const { return rw; }
  KS_FN int k_f0() const { return rw + nr; }
  KS_FN int k_f1() const { return rw + nr + 1; }
  KS_FN int k_tol() const { return rw + nr + 2; }
  KS_FN int k_meta() const { return rw + nr + 3; }
  KS_FN int k_hot_words() const { return rw + nr + 4; }
};

// LDS plan of the pack kernel (byte offsets into the dynamic shared segment), computed by the host.
struct LdsPlan {
  int total_bytes;
  int off_alloc;      // i64 [nr][iw*64]      allocatable, SoA
  int off_avail;      // u64 [iw*64]          offering availability cells
  int off_kv;         // u64 [n_kv][iw]       compact kv_has rows (only values of keys some instance type defines)
  int off_keymask;    // u64 [3][nk][iw]      key_undef | key_compl | key_neg
  int off_allocok;    // u64 [iw]
  int off_kvslot;     // u16 [rw*64]          dictionary (word,bit) -> compact kv row, 0xFFFF = none
  int off_tmpl;       // u64 [T][c_hot_words] template records (claim-shaped: prefiltered its, total 0, head +inf)
  int off_tmplcold;   // u64 [T][cold_words]
  int off_order;      // u32 key[cap] | ord[cap] | pos[cap]
  int order_cap;
  int off_closed;     // u64 [cap/64]   (BIG: [stage_words])
  int off_stage;      // u64 [stage_words] live-set staging of the BIG engine
  int stage_words;    // BIG: ceil(max_claims / 64), else 0
  int off_cache;      // u64 [32][c_hot_words] record cache
  int off_scratch;    // Scratch
  int off_dgov;       // i64 [n_dg][nr]       daemon overhead per group
  int off_dgits;      // u64 [n_dg][iw]       group membership
  int n_kv;
  // The compact consolidation sweep (ksolve_pack_sweep4): `waves` wavefronts of one workgroup share the read-only tables above and
  // the template records; each owns its Scratch, record cache, claim order and closed bitmap — those offsets are wave 0's, wave w's
  // lie w * wave_stride bytes further. 0 waves: the plan of a one-wavefront kernel.
  int waves, wave_stride;
  int off_shared_misc;   // u32 [4]: templates that survived the prefilter, written by wave 0 before the workgroup's barrier
  int off_topo, topo_bytes;   // the topology groups' descriptors and small mutable state (registered domains, per-domain counts, non-empty-domain
                              // counts) in LDS — Engine::topo_to_lds, the one-problem kernels only; 0 bytes: not planned (they stay in HBM)
};

// Topology groups (topologygroup.go:55-77), regular groups then inverse anti-affinity groups; group sets are bit masks of
// `words` u64 words.
constexpr int kMaxTopoWords = 16;
struct TopoView {
  int n_groups, dom_words, words;
  uint64_t inverse_mask[kMaxTopoWords], initially_active[kMaxTopoWords];
  const uint8_t* type;           // [G] 0 spread, 1 affinity, 2 anti-affinity
  const int32_t* key;            // [G] dictionary key, -1 = kubernetes.io/hostname (domains are the bins)
  const int8_t* key_slot;        // [G] row of the group's key in the per-claim domain masks (keys of one dictionary word), -1 = none
  int n_key_slots;
  int slot_key[4];               // dictionary key of each slot
  const int16_t* host_slot;       // [G] row of a hostname group in the per-bin counters, -1 for dictionary keys
  int n_host_groups;
  const int32_t *max_skew, *min_domains;
  const uint64_t* domains0;      // [G][dom_words] registered domains at creation
  const int32_t* counts0;        // [G][dom_words*64]
  const int32_t* node_counts0;   // [n_host_groups][n_nodes]
  const int32_t* nonzero0;       // [G] domains with a positive count at creation
  const uint8_t *f_affinity, *f_taint;   // [G] TopologyNodeFilter policies == Honor
  const uint32_t* f_first;       // [G+1]
  ReqTable f_reqs;
  const uint64_t* f_tolerates;   // [G]
  const uint16_t* value_rank;    // [req_words*64]
  const int32_t* node_host_value;  // [n_nodes]
  int n_alias;                    // classes of same-hash groups of which only the first created member exists
  const int16_t* alias_class;     // [G] class id or -1 (null when n_alias == 0)
  uint64_t alias_mask[kMaxTopoWords];  // groups with alias_class >= 0
  const uint64_t* cls_topo;      // [n_classes][2*words] owned | selected (class_gather)
  // resident clusters: where registered domains come from (a probe re-derives them without its removed nodes)
  const uint64_t* dom_universe;  // [G][dom_words] offered by NodePools / instance types; null = not a resident cluster
  const int32_t* dom_regs0;      // [G][dom_words*64] existing nodes that register the domain
};

struct ProblemView {
  Dict dict;
  int n_res, n_its, it_words;
  const int64_t* it_alloc;       // [n_res][n_its]
  const int64_t* it_cap;         // [n_res][n_its]
  const uint64_t* it_alloc_ok;   // [it_words] instance types whose allocatable has no negative dimension (resources.go:190)
  const uint64_t* it_off_avail;  // [n_its] bit zone*4+ct
  const double* it_off_price;    // [n_its][64]
  int n_zones, n_cts;
  // offering override groups (types.go:202-269): extra allocatable groups behind each type's base group. fits() passes a
  // type when some group holds the requests and has a compatible offering (nodeclaim.go:624-638).
  int n_xg;
  const uint32_t* xg_it;         // [n_xg] owning instance type
  const int64_t* xg_alloc;       // [n_res][n_xg]
  const uint64_t* xg_avail;      // [n_xg] offering cells of the group
  const uint64_t* it_base_avail; // [n_its] offering cells of the base group (== it_off_avail when n_xg == 0)
  int64_t xg_bonus[8];           // per dimension: the most an override group adds over its type's base allocatable (headroom bound)
  // instance-type requirement index, built by the prepass kernel (k_build_it_index):
  const uint64_t* kv_has;        // [req_words*64][it_words] ITs whose requirement on the value's key Has(value) (incl. complements)
  const uint64_t* key_undef;     // [n_keys][it_words] ITs that do not define the key
  const uint64_t* key_compl;     // [n_keys][it_words] ITs whose requirement on the key is a complement (NotIn/Exists)
  const uint64_t* key_neg;       // [n_keys][it_words] ITs whose operator on the key is NotIn or DoesNotExist
  ReqTable it_reqs;

  int n_templates;
  ReqTable tmpl_reqs;
  const uint64_t* tmpl_taints;
  const uint64_t* tmpl_its;      // [n_templates][it_words]
  const uint32_t* tmpl_limit_mask;
  const int64_t* tmpl_limits;    // [n_templates][n_res+1]
  // daemon-overhead groups (scheduler.go:963-1043): every template has at least one group; its groups partition its types
  int n_dg;
  const int* dg_first;           // [n_templates+1]
  const int64_t* dg_ov;          // [n_dg][n_res]
  const uint64_t* dg_its;        // [n_dg][it_words]
  uint64_t dg_nonzero;           // group has a non-zero overhead vector
  uint64_t dg_nonempty;          // group has at least one compatible daemon pod (addDaemonRequests, nodeclaim.go:353-377)
  // reserved offerings + ReservationManager (reservationmanager.go:28-110); reserved_on = feature gate and any reservation
  int reserved_on, reserved_strict, n_resv, key_rid, ct_reserved;
  const int32_t* resv_cap0;      // [n_resv]
  const uint32_t* it_resv_first; // [n_its+1]
  const uint8_t *resv_zone, *resv_id;
  const double* resv_price;

  // host ports (hostportusage.go): masks over the problem's distinct <ip, port, protocol> triples
  const uint64_t* node_removed;  // [node_words] existing nodes that are absent in this probe of a resident cluster (null: none)
  int hp_on;                     // some pod binds a host port
  const uint64_t* cls_hp;        // [n_classes][2] triples the class binds | triples that match one of them
  int vol_on;                    // some pod has volume requirement alternatives
  const uint64_t* cls_vol;       // [n_classes] first | count << 32 into vol_reqs
  ReqTable vol_reqs;             // the alternatives (ksolve_problem_desc.volume_reqs)
  const uint64_t* dg_hp;         // [n_dg] triples of each daemon-overhead group's daemon pods
  const uint64_t* node_hp0;      // [n_nodes] triples in use on each existing node before the solve (null = none)

  int n_pods, n_rows;
  const int32_t* row_next;       // [n_rows] relaxation chain
  const uint32_t* row_class;     // [n_rows] class id (k_classify)
  const uint8_t* pod_is_pending;

  int n_classes;
  const int64_t* cls_requests;   // [n_classes][n_res]
  ReqTable cls_reqs, cls_strict;
  const uint64_t* cls_tolerates; // [n_classes]
  const int64_t* min_request;    // [n_res] min over classes (for the closed-claim test)
  const uint64_t* cls_hot;       // [n_classes][k_hot_words]  (class_gather)
  const uint64_t* cls_cold;      // [n_classes][cold_words]
  RecLayout lay;
  LdsPlan lds;
  const uint16_t* kv_slot;       // [req_words*64] compact kv row of a dictionary value (host-built)
  uint32_t it_keys;              // keys that at least one instance type defines

  const uint32_t* sorted_pods;   // [n_pods] queue order (k_sort)

  // existing nodes in sortExistingNodes order (scheduler.go:845-858); SoA so that 64 lanes probe 64 nodes coalesced
  int n_nodes, node_words;       // node_words = ceil(n_nodes/64)
  const uint64_t* node_taints;   // [n_nodes]
  const uint8_t* node_flags;     // [n_nodes] bit0 initialized, bit1 under consolidateAfter
  const uint8_t* pod_from_deleting; // [n_pods]
  const int32_t* pod_node;       // [n_pods] resident clusters: the existing node a pod row is bound to, -1 = pending; null otherwise
  // CSI volume limits of existing nodes (VolumeUsage, volumeusage.go:178-209): volume = a distinct <driver, PVC> pair
  int pv_on, n_pv_drivers;       // pv_on: some node has a limit
  const uint8_t* pv_driver;      // [n_volumes]
  const uint32_t* pod_pv_first;  // [n_pods+1] CSR into pod_pvs
  const uint32_t* pod_pvs;
  const uint32_t* node_pv_first; // [n_nodes+1] CSR into node_pvs (ascending ids): the volumes in use before the solve
  const uint32_t* node_pvs;
  const int32_t* node_pv_limit;  // [n_nodes][n_pv_drivers], -1 = none
  // resident-cluster probes (ksolve_sweep, ksolve_probe_create): computed once per base handle, shared by every probe
  const uint64_t* n_dead0;       // [n_classes][node_words] class k cannot go on the PRISTINE node e for a reason that precedes topology (ksolve_node_dead0)
  const uint64_t* node_skip;     // [node_words] the nodes under consolidateAfter (node_flags bit1) as a bitmap
  const uint32_t* node_skip_prefix;   // [node_words + 1] how many of them precede each word
  TopoView topo;
  int big;                       // more in-flight claims than the LDS order holds: Engine<W, true, true> (order in HBM)
  int plain;                     // no topology groups, existing nodes, daemon overhead, minValues, reservations or bounds (any size)
  int strict_same;               // PodData.StrictRequirements == Requirements for every pod row (no preferred terms): one table uploaded, two gathered
  int plain_topo;                // the same, but for topology groups: what the spread engine (topo_engine.h) looks at
  int lite;                      // plain and small enough for the register tables:, existing nodes, daemon overhead, minValues or reservations: Engine<W, false>
};

struct Counters {
  unsigned long long bin_evaluations, full_evaluations, it_evaluations, queue_pops, sorts, slow_sorts, relaxations, column_resets;
  unsigned long long ref_bin_evaluations;  // V: candidate bins the reference would have evaluated
  unsigned long long cycles[24];           // shader clock spent per engine phase (profiling aid)
  unsigned long long full_filters;         // filterInstanceTypesByRequirements runs that had to re-evaluate compatibility + offerings
  unsigned long long node_block_steps;     // probes: 64-word steps over a class's n_dead0 row (512 B each) in the existing-node scan
  unsigned long long node_evaluations;     // existing nodes whose tables a scan actually read (the live bits of the blocks it stopped at)
};

struct Workspace {
  int max_claims, claim_words;   // claim_words = ceil(max_claims/64)
  // claims (AoS by claim; see RecLayout)
  uint64_t* c_hot;               // [max_claims][c_hot_words]
  uint64_t* c_cold;              // [max_claims][cold_words]
  uint64_t* c_keymask;           // [n_key_slots][max_claims] values each claim still admits on a topology key
  uint64_t* kv_claims;           // [n_key_slots][64][claim_words] inverse of c_keymask: the claims that admit value v (one word AND/OR
                                 //   per 64 claims in the scan prefilter instead of one load per claim)
  uint64_t* host_le;             // [n_host_groups][2][claim_words] claims whose per-claim counter of a hostname group is <= t-1 / <= t,
                                 //   t = maxSkew of a spread group, 0 of an anti-affinity group: the only two limits its pods ever test
  uint64_t* c_reserved;          // [max_claims] reservation ids held by each claim
  uint64_t* c_hp;                // [max_claims] host-port triples bound by the claim's pods (null unless hp_on)
  uint64_t* n_hp;                // [n_nodes] host-port triples in use on each existing node
  int64_t* c_headroom;           // [n_res][max_claims] SoA copy of the records' headroom: lane-per-claim prefilter of the scan
  // order (pdq_emul.h): lives in LDS while it fits (LdsPlan.order_cap), these are the HBM spill arrays
  uint32_t *o_key, *o_ord, *o_pos;
  // BIG engine (run_order.h): one ring per pod count; o_pos holds the claims' ring slots there, o_key / o_ord the array form
  uint32_t* o_ring;              // all rings
  uint32_t* o_cnt;               // [max_claims] pod count of the claim
  uint32_t* run_tabs;            // [3][run_kmax] head | size | prefix of every count's ring (the first kRunMaxCount entries live in LDS)
  const uint32_t* run_off;       // [run_kmax] first word of each count's ring
  const uint8_t* run_log;        // [run_kmax] log2 of its capacity
  int run_kmax;                  // pods of the problem + 2
  // first-fit pruning
  uint64_t* dead;                // [n_classes][claim_words] bit set = claim known infeasible for the class
  // existing nodes (mutable part): ExistingNode.requirements / remainingResources / Pods (existingnode.go:32-45)
  uint64_t* n_mask;              // [req_words][n_nodes]
  uint32_t *n_defined, *n_complement; // [n_nodes]
  uint32_t *n_hg, *n_hl;         // [n_nodes] keys with an integer bound on the node (nullptr: no pod of the problem carries Gt / Lt). Labels have none;
  int64_t *n_gte, *n_lte;        // [n_keys][n_nodes]   a Gt / Lt pod leaves one on a key a NotIn pod defined there (requirement.go:181-214)
  int64_t* n_remaining;          // [n_res][n_nodes]
  uint32_t* n_npods;             // [n_nodes]
  uint64_t* n_dead;              // [n_classes][node_words] node known infeasible for the class
  uint64_t* pv_log;              // [pods' volume entries] node << 32 | volume: what the pods placed on existing nodes in this solve added (VolumeUsage.Add)
  // pristine copies restored at the start of every solve
  const uint64_t* n_mask0; const uint32_t *n_defined0, *n_complement0; const int64_t* n_remaining0;
  // topology group state (TopologyGroup.domains / emptyDomains, topologygroup.go:74-76)
  uint64_t* tg_domains;          // [G][dom_words] registered domains (Record can add one, topologygroup.go:143-150)
  int32_t* tg_counts;            // [G][dom_words*64]
  int32_t* tg_node_counts;       // [n_host_groups][n_nodes]      hostname groups: pods per existing node (probes: [n_host_groups][ov_cap], what the probe adds to TopoView::node_counts0)
  int32_t* tg_claim_counts;      // [n_host_groups][max_claims]   hostname groups: pods per in-flight claim
  int32_t* tg_nonzero;           // [G] number of domains with a positive count
  int32_t* tg_alias_active;      // [n_alias] the member of each alias class that exists, -1 = none yet
  int32_t* tg_regs;              // [G][dom_words*64] probes of a resident cluster: nodes registering each domain without the removed ones (scratch of the probe's set-up)
  // queue (queue.go): circular buffer of pod ids + lastLen
  uint32_t* queue;               // [n_pods+1]
  uint32_t* last_len;            // [n_pods] 0 = never pushed
  // template state
  uint64_t* t_its;               // [n_templates][it_words] prefiltered instance types (scheduler.go:159)
  int64_t* t_remaining;          // [n_templates][n_res+1]
  // results
  int32_t* assign;               // [n_pods]
  uint8_t *err, *diag;           // [n_pods]
  uint32_t* slot;                // [n_pods]
  // ---- a probe of a resident cluster (disruption/helpers.go:53-155: Solve() on "the cluster without these nodes, with these
  // pods pending") ----  The cluster's node tables stay pristine and shared (n_*0, ProblemView::node_hp0); the probe keeps the
  // few nodes it commits pods to in an open-addressing overlay: the mutable n_* arrays above then have ov_cap slots (their
  // stride) instead of n_nodes, n_dead is not used (ProblemView::n_dead0 + the nodes in pr_revived), and the per-pod outputs
  // (assign, slot, err, diag, last_len) are indexed by the pod's position in pr_sorted.
  int probe;
  int pr_n_pods;                 // pods this probe schedules
  const uint32_t* pr_sorted;     // [pr_n_pods] their pod indices in the base problem, in queue order
  const uint32_t* pr_removed;    // [pr_n_removed] existing nodes that are not part of the simulation (helpers.go:76-80), ascending
  int pr_n_removed;
  const int64_t* pr_limits;      // [n_templates][n_res+1] NodePool limits with the removed nodes' capacity handed back, or null = the base problem's
  int pr_order_cap;              // claims the launch's LDS plan can order
  uint32_t* ov_key;              // [ov_cap] node + 1, 0 = free
  int ov_cap;                    // a power of two >= 2 * min(pr_n_pods, n_nodes)
  uint32_t* pr_revived;          // [ov_cap] overlaid nodes whose pristine rejections a commit has voided (revives_rejections)
  // scalars
  int* n_claims_out;
  int* status_out;               // 0 ok, 1 capacity exceeded, 2 cancelled/timed out
  volatile int* cancel_flag;
  long long max_steps;
  int min_values_best_effort;
  Counters* counters;
};

constexpr int kSweepLdsExtra = 4096;   // LDS behind the compact sweep's plan: the cluster's view once per workgroup + a workspace record per wavefront (ksolve_pack_sweep4)

struct BatchItem {  // one scheduling problem of a batched pack launch
  ProblemView pv;
  Workspace ws;
};

}  // namespace ks
