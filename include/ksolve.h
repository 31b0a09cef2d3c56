/*
 * ksolve.h — C ABI of the MI355X-native provisioning solver.
 *
 * This is the drop-in boundary for ONE path of kubernetes-sigs/karpenter: the provisioning scheduler's
 *   scheduling.NewScheduler(...)            pkg/controllers/provisioning/scheduling/scheduler.go:127-215
 *   (*Scheduler).Solve(ctx, pods)           pkg/controllers/provisioning/scheduling/scheduler.go:440-519
 * as called from Provisioner.Schedule (provisioner.go:359,430) and disruption.SimulateScheduling (helpers.go:113,128).
 * The reference is pure Go and has no FFI for this path; these are the entry points a cgo shim binds instead
 * (go/ksolve_shim.go). Plain pointers and sizes only: no Go pointers are retained after a call
 * returns, inputs are caller-owned and read-only for the duration of the call, outputs are library-owned and
 * released with ksolve_results_free.
 *
 * Data model ("KSP", flat SoA): label keys and values are dictionary-encoded by the caller. A requirement
 * (pkg/scheduling/requirement.go:36-43) on key k is {complement bit, u64 value-bitmask words over k's dictionary,
 * optional inclusive int bounds, optional minValues}; a requirement set (requirements.go:34) is one such slot per key.
 * Resources (pkg/utils/resources) are int64 in a per-dimension scale chosen by the caller so every quantity is exact.
 */
#ifndef KSOLVE_H
#define KSOLVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSOLVE_ABI_VERSION 8
#define KSOLVE_MAX_KEYS 32        /* requirement keys per problem (one bit each in the u32 flag words) */
#define KSOLVE_MAX_RES 8          /* resource dimensions */
#define KSOLVE_MAX_TEMPLATES 32   /* NodeClaimTemplates (NodePools that survived prefiltering) */
#define KSOLVE_MAX_ITWORDS 32     /* ceil(n_instance_types / 64) */
#define KSOLVE_MAX_ZONES 16       /* distinct offering zones  */
#define KSOLVE_MAX_CAPTYPES 4     /* distinct offering capacity types */
#define KSOLVE_MAX_VOLUME_DRIVERS 8
#define KSOLVE_MAX_OVERRIDE_GROUPS 4096 /* extra allocatable groups (offering overrides) per problem */
#define KSOLVE_MAX_TOPO_GROUPS 1024 /* topology groups per problem; pod group masks take ceil(n/64) words */

typedef enum {
  KSOLVE_OK = 0,
  KSOLVE_ERR_INVALID = 1,      /* malformed problem description */
  KSOLVE_ERR_UNSUPPORTED = 2,  /* valid Karpenter input outside what this build solves on the device (never a CPU fallback) */
  KSOLVE_ERR_DEVICE = 3,       /* HIP runtime failure; see ksolve_last_error */
  KSOLVE_ERR_NO_DEVICE = 4,    /* no gfx950 device / HIP runtime not usable */
  KSOLVE_ERR_CANCELLED = 5,    /* deadline or ksolve_cancel: results are partial but valid (scheduler.go:477-480,:518) */
  KSOLVE_ERR_CAPACITY = 6      /* a fixed device-side capacity (claims) was exceeded */
} ksolve_status;

/* Per-pod error codes (Results.PodErrors, scheduler.go:284). */
typedef enum {
  KSOLVE_POD_OK = 0,
  KSOLVE_POD_TAINTS = 1,          /* did not tolerate taint                         taints.go:91 */
  KSOLVE_POD_INCOMPATIBLE = 2,    /* incompatible requirements                      nodeclaim.go:134 */
  KSOLVE_POD_TOPOLOGY = 3,        /* unsatisfiable topology constraint              topology.go:241 */
  KSOLVE_POD_INSTANCE_TYPES = 4,  /* InstanceTypeFilterError (+6 diagnostic bits)   nodeclaim.go:437-459 */
  KSOLVE_POD_RESOURCES = 5,       /* exceeds node resources                         existingnode.go:97 */
  KSOLVE_POD_NO_TEMPLATES = 6,    /* nodepool requirements filtered out all types   scheduler.go:605 */
  KSOLVE_POD_LIMITS = 7,          /* nodepool limits                                scheduler.go:713,718 */
  KSOLVE_POD_RESERVED = 8,        /* ReservedOfferingError                          nodeclaim.go:341,346 */
  KSOLVE_POD_EXISTING = 9,
  KSOLVE_POD_MIN_VALUES = 10      /* minValues requirement is not met               types.go:430 */
} ksolve_pod_error;

/* A table of requirement sets, SoA over `n` entities. words_per_set = problem.req_words (sum over keys of the
 * key's dictionary words). mask[e * req_words + key_word_off[k] + w]. Flag words carry one bit per key. */
typedef struct {
  uint32_t n;
  const uint64_t* mask;       /* n * req_words */
  const uint32_t* defined;    /* n : key present in the set (requirements.go Has) */
  const uint32_t* complement; /* n : requirement.complement */
  const uint32_t* has_gte;    /* n */
  const uint32_t* has_lte;    /* n */
  const int64_t* gte;         /* n * n_keys, may be NULL when no entity has bounds */
  const int64_t* lte;         /* n * n_keys, may be NULL */
  const int32_t* min_values;  /* n * n_keys, -1 = nil; may be NULL */
} ksolve_reqsets;

/* Topology groups (topologygroup.go:55-77): Topology.topologyGroups then Topology.inverseTopologyGroups
 * (topology.go:52-56), pre-deduplicated by the caller with TopologyGroup.Hash() semantics (topologygroup.go:188-222).
 * A group on a dictionary key counts pods per dictionary value of that key; a kubernetes.io/hostname group (key = -1)
 * counts pods per bin (existing node / in-flight NodeClaim) — the solver owns those domains. */
typedef struct {
  uint32_t n;                  /* <= KSOLVE_MAX_TOPO_GROUPS */
  const uint8_t* type;         /* 0 spread, 1 pod affinity, 2 pod anti-affinity (topologygroup.go:35-41) */
  const uint8_t* inverse;      /* 1 = member of Topology.inverseTopologyGroups (topology.go:56) */
  const uint8_t* initially_active; /* 1 = created by NewTopology (topology.go:68-103); 0 = created when a pod relaxes into a
                                    * variant that owns it (topology.go:162-194): it sees no Record before that moment */
  const int32_t* key;          /* requirement key index, or -1 for kubernetes.io/hostname */
  const int32_t* max_skew;
  const int32_t* min_domains;  /* -1 = nil */
  uint32_t domain_words;       /* mask words per group in `domains` / 64 counters each in `init_counts` */
  const uint64_t* domains;     /* n * domain_words : registered domains of a dictionary key (topologygroup.go:103-107) */
  const int32_t* init_counts;  /* n * domain_words * 64 : pods pre-counted per domain value (topology.go:361-459) */
  const int32_t* init_node_counts; /* n * n_nodes : hostname groups, pods pre-counted per existing node; may be NULL */
  const uint8_t* filter_affinity_honor; /* TopologyNodeFilter: nodeAffinityPolicy == Honor (topologynodefilter.go:68-79) */
  const uint8_t* filter_taint_honor;    /* nodeTaintsPolicy == Honor */
  const uint32_t* filter_first;         /* n+1 : CSR into filter_reqs (TopologyNodeFilter.Requirements, OR'd) */
  ksolve_reqsets filter_reqs;
  const uint64_t* filter_tolerates;     /* n : distinct-taint mask the creating pod tolerates */
  const uint16_t* value_rank;  /* req_words*64 : rank of a dictionary value's name among its key's values (Go map iteration
                                * order is undefined where the reference breaks ties, topologygroup.go:259,355,372; ties go
                                * to the lexicographically smallest domain) */
  const int32_t* node_hostname_value;   /* n_nodes : value index of the node's hostname in key_hostname's dictionary, -1 = none */
  const int32_t* alias_class;  /* n, may be NULL; -1 = none. Groups created by relaxation (initially_active == 0) that share one
                                * TopologyGroup.Hash() but were built from different pods (the hash ignores the node filter's
                                * values and the domains, topologygroup.go:188-222) carry the same class id in
                                * [0, n_alias_classes): Topology.Update reuses the group found by hash (topology.go:162-194), so
                                * only the member whose owner relaxes FIRST comes to exist, and every later owner of another
                                * member owns that one instead. */
  uint32_t n_alias_classes;
  /* Resident clusters (ksolve_problem_desc.pod_node != NULL; ksolve_sweep / ksolve_probe_create): init_counts then count EVERY
   * bound pod of the cluster, also those that are pod rows, and a probe takes the share of its displaced pods and removed nodes
   * out again (what countDomains does by excluding the pods being scheduled, topology.go:92-94, :361-459). For that it needs to
   * know which registered domains only exist through nodes: */
  const uint64_t* domain_universe;   /* n * domain_words : domains the NodePools / instance types offer (ForEachDomain, topologydomaingroup.go:61-72); NULL = `domains` */
  const int32_t* domain_node_regs;   /* n * domain_words * 64 : existing nodes that pass the group's node filter and carry the domain (topology.go:376-384); may be NULL */
} ksolve_topology;

typedef struct {
  uint32_t abi_version;

  /* ---- dictionaries ---- */
  uint32_t n_keys;                 /* <= KSOLVE_MAX_KEYS */
  const uint32_t* key_word_off;    /* n_keys+1 : first mask word of each key; req_words = key_word_off[n_keys] */
  uint32_t well_known_mask;        /* v1.WellKnownLabels as key bits (labels.go:75-84 plus provider additions) */
  int32_t key_instance_type;       /* key index of node.kubernetes.io/instance-type; its dictionary IS the instance type list */
  int32_t key_zone, key_capacity_type;
  int32_t key_hostname;            /* key index of kubernetes.io/hostname when some pod/node requirement mentions it, else -1 */
  const int64_t* value_int;        /* req_words*64 : strconv.Atoi(value) */
  const uint64_t* value_is_int;    /* req_words : bit set when the value parses as an integer (requirement.go:339) */

  /* ---- resources ---- */
  uint32_t n_res;                  /* dims; 0 = cpu, 1 = memory (queue.go:72-90 sorts on these two) */

  /* ---- instance types: cloudprovider.InstanceType (types.go:123-142) ---- */
  uint32_t n_its;
  const int64_t* it_allocatable;   /* n_res * n_its (SoA): Allocatable() = capacity - overhead  (types.go:271-294) */
  const int64_t* it_capacity;      /* n_res * n_its */
  ksolve_reqsets it_reqs;          /* InstanceType.Requirements */
  const uint64_t* it_offering_avail; /* n_its : bit (zone_idx*4 + ct_idx) set for an Available offering (types.go:470-486) */
  const double* it_offering_price;   /* n_its * 64 */
  uint32_t n_zones, n_captypes;    /* zone_idx / ct_idx are value indices in key_zone / key_capacity_type dictionaries */

  /* ---- offering capacity / overhead overrides (types.go:202-269): an instance type whose AVAILABLE offerings carry a
   *      CapacityOverride / OverheadOverride has one more AllocatableOfferings group per distinct override pair, behind the
   *      base group (it_allocatable + the offerings without overrides). fits() accepts the type when SOME group both holds
   *      the requests and has an offering compatible with the requirements (nodeclaim.go:624-638). it_offering_avail /
   *      it_offering_price stay the union over all groups (hasOffering, prices). n_override_groups == 0: nothing below is read. ---- */
  uint32_t n_override_groups;            /* extra groups over all instance types, <= KSOLVE_MAX_OVERRIDE_GROUPS */
  const uint32_t* override_it;           /* n_override_groups : the instance type the group belongs to */
  const int64_t* override_allocatable;   /* n_res * n_override_groups (SoA) : computeAllocatable(override) (types.go:271-294) */
  const uint64_t* override_avail;        /* n_override_groups : cells (zone_idx*4 + ct_idx) of the group's available offerings */
  const uint64_t* it_base_avail;         /* n_its : cells of the available offerings WITHOUT overrides (the base group) */

  /* ---- templates: NodeClaimTemplate per NodePool, in OrderByWeight order (nodepool.go:161-171) ---- */
  uint32_t n_templates;
  ksolve_reqsets tmpl_reqs;        /* nodeclaimtemplate.go:66-94 */
  const uint64_t* tmpl_taints;     /* n_templates : mask over the problem's distinct taints */
  const uint64_t* tmpl_its;        /* n_templates * it_words : instance types offered by the pool (GetInstanceTypes) */
  const uint32_t* tmpl_limit_mask; /* n_templates : bit r set => limits[r] applies; bit n_res => "nodes" limit */
  const int64_t* tmpl_limits;      /* n_templates * (n_res+1) : remaining = limits - existing capacity (scheduler.go:183,835) */

  /* ---- daemonset overhead (scheduler.go:963-1043): the instance types of a template are partitioned into groups that
   *      share the same set of compatible daemonset pods; a type must fit requests + its group's overhead
   *      (nodeclaim.go:558-566) and FinalizeScheduling adds the smallest overhead to the claim's requests
   *      (nodeclaim.go:353-377). NULL tmpl_daemon_first = no daemonsets. At most 64 groups per problem. ---- */
  const uint32_t* tmpl_daemon_first;     /* n_templates+1 : CSR into the group arrays */
  const uint64_t* daemon_group_its;      /* n_groups * it_words */
  const int64_t* daemon_group_overhead;  /* n_groups * n_res, incl. pods = number of daemon pods */
  const uint8_t* daemon_group_nonempty;  /* n_groups : at least one daemon pod is compatible with the group */

  /* ---- reserved offerings (capacity type "reserved", types.go:470-486) and the ReservationManager
   *      (reservationmanager.go:28-110). Only consulted when ksolve_options.reserved_capacity is set. ---- */
  uint32_t n_reservations;               /* distinct reservation ids, <= 64 */
  const int32_t* reservation_capacity;   /* n_reservations : smallest ReservationCapacity over the id's offerings (reservationmanager.go:45-60) */
  int32_t key_reservation_id;            /* key index of cloudprovider.ReservationIDLabel; value index == reservation index; -1 = none */
  int32_t captype_reserved;              /* value index of "reserved" in key_capacity_type's dictionary, -1 = none */
  const uint32_t* it_reserved_first;     /* n_its+1 : CSR over the AVAILABLE reserved offerings of each instance type */
  const uint8_t* reserved_zone;          /* per reserved offering: zone value index */
  const uint8_t* reserved_id;            /* per reserved offering: reservation index */
  const double* reserved_price;          /* per reserved offering */

  /* ---- pods (one row per pod *variant*: row p < n_pods is the pod as submitted; rows >= n_pods are the
   *      pre-computed results of Preferences.Relax (preferences.go:38-57), chained through pod_next_variant) ---- */
  uint32_t n_pods;
  uint32_t n_pod_rows;
  const int64_t* pod_requests;     /* n_res * n_pod_rows (SoA) : RequestsForPods incl. pods=1 (resources.go:30-38) */
  ksolve_reqsets pod_reqs;         /* PodData.Requirements  (scheduler.go:554-580) */
  ksolve_reqsets pod_strict_reqs;  /* PodData.StrictRequirements; mask may alias pod_reqs when identical */
  const uint64_t* pod_tolerates;   /* n_pod_rows : bit i set when the pod tolerates distinct taint i (taints.go:83-95) */
  const int32_t* pod_next_variant; /* n_pod_rows : row of the next relaxation, -1 = none */
  const uint64_t* pod_topo_owned;  /* n_pod_rows * topo_words, topo_words = ceil(topo.n / 64): topology groups the pod variant
                                    * owns (topology.go:187, :352); NULL when topo.n == 0 */
  const uint64_t* pod_topo_selected; /* n_pod_rows * topo_words : groups whose namespaces + selector match the pod (topologygroup.go:442) */
  const int64_t* pod_creation;     /* n_pods : CreationTimestamp seconds */
  const uint64_t* pod_uid_hi;      /* n_pods : (hi,lo) compare like the UID strings (queue.go:107) */
  const uint64_t* pod_uid_lo;
  const uint8_t* pod_is_pending;   /* n_pods : Status.Phase == Pending (scheduler.go:628) */
  const uint8_t* pod_from_deleting_node; /* n_pods */
  const int32_t* pod_node;         /* n_pods or NULL. Non-NULL marks a RESIDENT CLUSTER: the pod rows include the pods bound to the existing
                                    * nodes (pod_node = the node's index, -1 = pending), so that any set of nodes can be a probe's candidates
                                    * (ksolve_sweep); such a problem is only solved through probes */

  /* ---- host ports (hostportusage.go:39-117): bit masks over the problem's distinct <hostIP, hostPort, protocol> triples
   *      (<= 64). A pod joins a bin only when none of its triples matches (same protocol and port, equal IPs or one of them
   *      unspecified) a triple in use there: an existing node's (existingnode.go:87-93), or — per daemon-overhead group —
   *      the group's daemon pods' plus the pods already on the NodeClaim (nodeclaim.go:256-259, :562-565).
   *      pod_host_ports == NULL: no pod of the problem binds a host port and nothing below is read. ---- */
  const uint64_t* pod_host_ports;          /* n_pod_rows : triples the pod binds */
  const uint64_t* pod_host_port_conflicts; /* n_pod_rows : every triple of the problem that matches one of the pod's */
  const uint64_t* node_host_ports;         /* n_nodes : triples in use on the node (StateNode.HostPortUsage); may be NULL */
  const uint64_t* daemon_group_host_ports; /* n_groups : triples of the group's daemon pods (scheduler.go:990-993); may be NULL */

  /* ---- volume requirement alternatives (PodData.VolumeRequirements = volumeReqsByPod[pod.UID], scheduler.go:138, :222,
   *      :572): NodeClaim.CanAdd (nodeclaim.go:138-157, tryVolumeAlternative :164-242) and ExistingNode.CanAdd
   *      (existingnode.go:108-139, :143-168) intersect the bin's requirements — not the pod's — with the first alternative
   *      that passes every later check, trying them in order; the error kept is the last alternative's. Pods with equal lists
   *      carry equal (first, count) (it is part of the pod's class identity); every variant row of a pod carries the pod's.
   *      pod_volume_first == NULL: no pod of the problem has volume requirements and nothing below is read. ---- */
  uint32_t n_volume_reqs;                  /* requirement sets in volume_reqs */
  ksolve_reqsets volume_reqs;              /* the alternatives of every distinct list, list after list; no minValues */
  const uint32_t* pod_volume_first;        /* n_pod_rows : first alternative of the pod's list */
  const uint32_t* pod_volume_count;        /* n_pod_rows : alternatives of the pod (0 = none: one pass with no extra requirement) */

  /* ---- existing nodes, already in sortExistingNodes order (scheduler.go:845-858) ---- */
  uint32_t n_nodes;
  ksolve_reqsets node_reqs;        /* labels + hostname (existingnode.go:66-70); hostname value index in node_hostname */
  const uint64_t* node_taints;
  const int64_t* node_remaining;   /* n_res * n_nodes : Available - daemon overhead (existingnode.go:47-64) */
  const uint8_t* node_initialized;
  const uint8_t* node_under_consolidate_after;

  ksolve_topology topo;

  /* ---- distinct taints of the problem (for filter_taint_honor only the masks matter) ---- */
  uint32_t n_taints;

  /* ---- CSI volume limits of existing nodes: StateNode.VolumeUsage() (statenode.go:411, :466-490), VolumeUsage.ExceedsLimits /
   *      Add (pkg/scheduling/volumeusage.go:193-209), checked by ExistingNode.CanAdd right after the taints (existingnode.go:88)
   *      and updated by ExistingNode.Add (:179). A volume is a distinct <CSI driver, PVC> pair of the problem (what
   *      scheduling.GetVolumes resolves per pod, volumeusage.go:83-114 — upstream of Solve()); volume ids are numbered so that
   *      volume_driver[id] is its driver. A pod joins a node only if, for every driver with a limit on that node, the union of
   *      the node's and the pod's volumes of the driver stays within the limit. A node that is over a limit before the solve
   *      rejects every pod (ExceedsLimits walks the union's drivers): the flattener gives such a node negative remaining
   *      resources. New NodeClaims have no limits (no CSINode yet). n_volume_drivers == 0: nothing below is read. ---- */
  uint32_t n_volume_drivers;               /* drivers that have a limit on some node, <= KSOLVE_MAX_VOLUME_DRIVERS */
  uint32_t n_volumes;                      /* distinct volumes of those drivers */
  const uint8_t* volume_driver;            /* n_volumes */
  const uint32_t* pod_pv_first;            /* n_pods + 1 : CSR into pod_pvs (variant rows share their pod's) */
  const uint32_t* pod_pvs;                 /* volume ids, distinct within a pod */
  const uint32_t* node_pv_first;           /* n_nodes + 1 : CSR into node_pvs */
  const uint32_t* node_pvs;                /* volume ids in use on the node, ascending */
  const int32_t* node_pv_limit;            /* n_nodes * n_volume_drivers : attach limit, -1 = none */
} ksolve_problem_desc;

typedef struct {
  uint32_t min_values_best_effort; /* MinValuesPolicyBestEffort (scheduler.go:117) */
  uint32_t max_claims;             /* device capacity for in-flight NodeClaims; 0 = n_pods */
  int64_t max_steps;               /* stand-in for the ctx deadline: stop after this many queue pops, -1 = none */
  uint32_t device;                 /* HIP device ordinal */
  uint32_t lds_claim_cap;          /* 0 = as many in-flight claims as the CU's LDS can order (<= 8192); smaller values shrink the
                                    * LDS footprint (more problems per CU in ksolve_solve_batch). A solve that needs more claims
                                    * is re-run on the engine variant that keeps the claim order in HBM. */
  uint32_t truncate_instance_types; /* > 0: Results.TruncateInstanceTypes(n) (scheduler.go:419-437; the provisioner passes 600):
                                    * every NodeClaim's instance types ordered by price (OrderByPrice, types.go:336-355, Go's unstable
                                    * sort.Slice reproduced) and capped at n; a claim whose capped list breaks minValues is reported in
                                    * ksolve_claims.truncation_failed */
  uint32_t reserved_capacity;      /* FeatureGates.ReservedCapacity (nodeclaim.go:308) */
  uint32_t reserved_offering_strict; /* DisableReservedCapacityFallback / ReservedOfferingModeStrict (scheduler.go:103, nodeclaim.go:339-348) */
  uint32_t engine;                 /* which pack engine may run. 0 = automatic: the cursor engine (csrc/fast_engine.h) for problems whose
                                    * requirement algebra is purely positive (In sets only, no topology / existing nodes / minValues /
                                    * reservations / daemon overhead), the general engine otherwise or whenever the cursor engine stops;
                                    * 1 = general engine only; 2 = cursor engine only (KSOLVE_ERR_UNSUPPORTED instead of the fallback:
                                    * tests use it to prove which engine produced a result); 3 = cursor engine only, with the claims' state
                                    * in HBM from the start (the plan the library moves to by itself when the LDS plan runs out of
                                    * claims); 4 = cursor engine only, claim state AND claim order in HBM from the start (the plan above
                                    * ~15,000 in-flight NodeClaims, up to 65,472); 5 = cursor engine only, LDS plan with one row of class slots on
                                    * TWO wavefronts (ksolve_pack_fast2: a second wavefront recomputes a NodeClaim's acceptance words
                                    * while the first places the next pod; kept for measurements — on the MI355X it is 5.7% slower than
                                    * the one-wavefront kernel every other setting runs, profiles/round5/pass_i);
                                    * 6 = spread engine only (csrc/topo_engine.h, round 6): the cursor engine's shape plus topology spread /
                                    * pod affinity on dictionary keys and spread / anti-affinity on the hostname (BASELINE configs[2]);
                                    * KSOLVE_ERR_UNSUPPORTED when the problem is outside that shape. Automatic (0) tries it first on every
                                    * problem that is plain but for its topology groups and falls back to the general engine when it
                                    * declines. All give identical Results. */
} ksolve_options;

/* One NodeClaim of Results.NewNodeClaims (scheduler.go:282, nodeclaim.go:43-62), in the order the reference's
 * s.newNodeClaims slice ends in. */
typedef struct {
  uint32_t n_claims;
  uint32_t it_words, req_words, n_keys, n_res;
  const int32_t* template_idx;   /* n_claims */
  const uint32_t* pod_count;     /* n_claims */
  const uint64_t* it_mask;       /* n_claims * it_words : InstanceTypeOptions */
  const int64_t* requests;       /* n_claims * n_res : Spec.Resources.Requests */
  const uint64_t* req_mask;      /* n_claims * req_words : Requirements after FinalizeScheduling */
  const uint32_t* req_defined, *req_complement, *req_has_gte, *req_has_lte;
  const int64_t* req_gte, *req_lte;     /* n_claims * n_keys */
  const int32_t* req_min_values;        /* n_claims * n_keys */
  const uint8_t* min_values_relaxed;    /* n_claims : annotation nodeclaim-min-values-relaxed (scheduler.go:763-772) */
  const double* cheapest_price;         /* n_claims : cheapest compatible available offering over InstanceTypeOptions */
  const uint32_t* hostname_seq;         /* n_claims : N of hostname-placeholder-%04d (nodeclaim.go:93) */
  const int32_t* ordered_instance_types; /* NULL unless options.truncate_instance_types: n_claims * n_instance_types, the claim's types in
                                          * OrderByPrice order; the first ordered_count[c] entries are the (truncated) options */
  const uint32_t* ordered_count;        /* n_claims */
  const uint8_t* truncation_failed;     /* n_claims : minValues no longer met after truncation (types.go:437-449) */
  uint32_t n_instance_types;
  const uint64_t* reserved_mask;        /* n_claims : reservation ids held by the claim (NodeClaim.reservedOfferings, nodeclaim.go:60) */
} ksolve_claims;

typedef struct {
  ksolve_status status;
  uint32_t n_pods;
  const int32_t* pod_assignment;   /* n_pods : >=0 index into claims; <= -2 existing node (-2 - node index); -1 unscheduled */
  const uint8_t* pod_error;        /* n_pods : ksolve_pod_error */
  const uint8_t* pod_error_diag;   /* n_pods : InstanceTypeFilterError bits (requirementsMet|fits<<1|hasOffering<<2|...) */
  const uint32_t* pod_slot;        /* n_pods : position of the pod inside its claim's / node's Pods slice */
  ksolve_claims claims;
  /* counters: V = candidate-bin evaluations, plus timing of the device phases in microseconds */
  uint64_t bin_evaluations, it_evaluations, queue_pops, sorts, slow_sorts, relaxations;
  uint64_t ref_bin_evaluations;    /* V as the reference algorithm would count it: every claim up to the accepting one */
  uint64_t phase_cycles[24];       /* profiling builds only (-DKSOLVE_PHASE_TIMERS), zero otherwise: shader clocks per pack-engine phase
                                    * (queue, class fetch, sort, scan, record load, CanAdd, commit, new claim, dead mark, trySchedule,
                                    * total, CanAdd and scan sub-phases) and a few diagnostic counts */
  double us_upload, us_prepass, us_pack, us_finalize, us_download;
  double packing_cost;
  uint32_t engine_used;            /* 1 = general engine, 2 = cursor engine, 3 = spread engine */
  uint32_t engine_fallback_reason; /* non-zero: why the cursor / spread engine handed the problem to the general engine (csrc/fast_engine.h
                                    * setup(): 1-8; run time: 20-28; csrc/topo_engine.h setup_topo(): 40-51, run time: 60-62) */
  uint32_t cursor_wide;            /* engine_used == 2: the memory plan it ran with. 0 = claim state and order in LDS (~3,000 in-flight
                                    * NodeClaims); 1 = the claims' state in HBM (~15,000); 2 = their order too (65,472) */
  uint32_t cursor_attempts;        /* runs of the cursor engine this solve took: 1, or one more per plan it outgrew (a later solve of the
                                    * handle starts with the plan that held) */
  void* impl;
} ksolve_results;

typedef struct ksolve_handle ksolve_handle;

/* One probe of a RESIDENT cluster — disruption.SimulateScheduling (disruption/helpers.go:53-155) runs Solve() on "the
 * cluster without these candidate nodes, with their pods pending", once per candidate set of a consolidation sweep
 * (singlenodeconsolidation.go:55-126, multinodeconsolidation.go:117-207). The base handle is created ONCE for the whole
 * cluster: every node as an existing node, every pod that some probe may have to place as a pod row. A probe names the
 * nodes that are not there and the pods to place; ksolve_probe_create makes a handle that SHARES the base handle's device
 * tables (dictionaries, instance types, templates, pod classes, queue order, pristine node state) and owns only its
 * workspace, so a sweep costs one upload + one classing pass, and its probes go to ksolve_solve_batch in ONE launch. */
typedef struct {
  const uint64_t* removed_nodes;   /* ceil(n_nodes / 64) words over the base problem's existing nodes: bit set = not in this simulation */
  uint32_t n_pods;
  const uint32_t* pods;            /* n_pods distinct pod indices (< base n_pods): the pods this simulation schedules */
  const int64_t* tmpl_limits;      /* n_templates * (n_res+1), or NULL = the base problem's: NodePool limits left once the capacity
                                    * of the removed nodes is handed back (scheduler.go:835-842) */
} ksolve_probe;

/* Validates and uploads a problem: device buffers + a HIP stream owned by the handle (NewScheduler). */
ksolve_status ksolve_create(const ksolve_problem_desc* desc, const ksolve_options* opts, ksolve_handle** out);
/* Runs Solve() on the device. One in-flight solve per handle; distinct handles are independent and thread-safe. */
ksolve_status ksolve_solve(ksolve_handle* h, ksolve_results* out);
/* Solves n independent problems with ONE launch of the pack kernel per device: block b is the wavefront of problem b. The
 * handles may live on several devices (ksolve_options.device): every device's problems run as one batch on that device, the
 * devices side by side (one host thread and one stream set each), and the call returns when all are done — the one-process
 * form of "partition the batch across the GPUs". Results are identical to n ksolve_solve calls; outs[i].status carries each
 * problem's status.
 * This is the entry point for consolidation sweeps (disruption/helpers.go:53-155 runs one Solve() per candidate set)
 * and for NodePool components of one provisioning pass. */
ksolve_status ksolve_solve_batch(ksolve_handle** handles, uint32_t n, ksolve_results* outs);
/* A handle for one probe of `base` (see ksolve_probe). `base` must outlive it and must not be solved concurrently with it;
 * results use the base problem's pod and node numbering (pods outside the probe: assignment -1, error 0; removed nodes take
 * no pods). Works with ksolve_solve, ksolve_solve_batch, ksolve_cancel, ksolve_destroy. A base problem with topology groups
 * must be a RESIDENT CLUSTER (ksolve_problem_desc.pod_node: the domain counts include every bound pod row, the domain universe
 * and the per-domain node registrations are given apart): a probe then takes its candidates' share out of its own copy of the
 * counters (topology.go:68-103, :310-355, :361-459). KSOLVE_ERR_UNSUPPORTED for topology groups on any other base. */
ksolve_status ksolve_probe_create(ksolve_handle* base, const ksolve_probe* probe, ksolve_handle** out);
/* A whole consolidation sweep in ONE call: n_probes simulations of the resident cluster `base` — single-node consolidation
 * tries every candidate (singlenodeconsolidation.go:55-126), multi-node consolidation every prefix its binary search can reach
 * (multinodeconsolidation.go:117-207), the validator replays commands (validation.go:297-357); each of them is
 * SimulateScheduling (disruption/helpers.go:53-155): Solve() on the cluster without the probe's nodes, with the probe's pods
 * pending. The descriptors are CSR arrays (one upload), every probe is one wavefront of one launch, its workspace is a slice
 * of one arena that the base handle keeps between calls (a probe owns only the claims it creates and an overlay of the few
 * nodes it commits pods to; the cluster's tables stay shared and pristine), and the results come back in one download. */
typedef struct {
  uint32_t n_probes;
  const uint32_t* node_off;        /* n_probes + 1 : CSR into nodes */
  const uint32_t* nodes;           /* existing-node indices of the base problem that are not part of the simulation (the candidates);
                                    * distinct within a probe (KSOLVE_ERR_INVALID otherwise), offsets non-decreasing */
  const uint32_t* pod_off;         /* n_probes + 1 : CSR into pods */
  const uint32_t* pods;            /* pod indices of the base problem the simulation schedules, distinct within a probe, any order.
                                    * Topology bases: EVERY bound pod row of a removed node must be listed — a hostname group's
                                    * per-node counter leaves the simulation with the node (the reference's countDomains would
                                    * still count a bound pod the caller chose not to reschedule, e.g. one a PodDisruptionBudget
                                    * holds back, topology.go:361-459 / helpers.go:86-95: keep such a node out of the sweep) */
  const int64_t* tmpl_limits;      /* NULL = the base problem's, else n_probes * n_templates * (n_res+1): NodePool limits with the removed
                                    * nodes' capacity handed back (scheduler.go:835-842) */
} ksolve_sweep_desc;

typedef struct {
  uint32_t n_probes;
  const int32_t* status;           /* n_probes : ksolve_status of each simulation (KSOLVE_ERR_CAPACITY: more NodeClaims than a probe may hold) */
  /* per pod, aligned with ksolve_sweep_desc.pods (same CSR offsets, same order as given) */
  const int32_t* pod_assignment;   /* >= 0: index into THIS PROBE's claims (claim_off[p] + value is the row in `claims`); <= -2 existing node
                                    * (-2 - node index); -1 unscheduled */
  const uint8_t* pod_error;        /* ksolve_pod_error */
  const uint8_t* pod_error_diag;
  const uint32_t* pod_slot;        /* position of the pod inside its claim's / node's Pods slice (node slots count only this probe's pods) */
  const uint32_t* claim_off;       /* n_probes + 1 : the probe's rows in `claims`, in the order the reference's s.newNodeClaims ends in */
  ksolve_claims claims;            /* every probe's NodeClaims, concatenated */
  const uint64_t* ref_bin_evaluations;   /* n_probes : V of each simulation */
  double us_upload, us_pack, us_finalize, us_download;
  /* what the launch(es) read, summed over the probes — the terms of the sweep kernel's algorithmic bytes */
  uint64_t total_bin_evaluations;        /* CanAdd calls: existing nodes, in-flight claims, new claims */
  uint64_t total_node_evaluations;       /* existing nodes whose tables a scan read */
  uint64_t total_node_block_steps;       /* 512-byte steps over the per-class rejection rows of the pristine nodes */
  uint32_t n_classes;                    /* pod classes of the base problem (rows of the per-class rejection table) */
  uint32_t it_words, n_nodes;
  double us_node_dead0;                  /* the once-per-cluster kernel that fills that table (every class x every pristine node) */
  void* impl;
} ksolve_sweep_results;

/* Runs the sweep. One sweep (or solve) per base handle at a time. Topology groups: see ksolve_probe_create (resident-cluster
 * bases only). Sweeps whose probes need more workspace than the arena budget (KSOLVE_SWEEP_ARENA_MB, default 4096) run as
 * several launches inside the call, cut by the probes' measured workspace sizes; a launch whose arena the device refuses is
 * retried at half its size. The function's status is that of the call (arguments, device); each simulation's own
 * status is in results.status. */
ksolve_status ksolve_sweep(ksolve_handle* base, const ksolve_sweep_desc* desc, ksolve_sweep_results* out);
/* The same sweep over several devices of the node in one call: `bases` are handles created from ONE resident-cluster problem with
 * different ksolve_options.device (replicas; nothing is exchanged between devices). The probes are cut into contiguous shares of
 * about equal displaced-pod counts, every device runs its share (one host thread, one or several launches each), the results
 * come back in probe order exactly as ksolve_sweep(bases[0], ...) would have produced them; the timings are those of the slowest
 * device. For a one-process caller that owns all GPUs of the node (BASELINE configs[4] on 8 x MI355X without a collective). */
ksolve_status ksolve_sweep_replicas(ksolve_handle** bases, uint32_t n_bases, const ksolve_sweep_desc* desc, ksolve_sweep_results* out);
void ksolve_sweep_results_free(ksolve_sweep_results* r);
/* The north_star's global packing summary: per instance type, how many of the NodeClaims of `results` launch on it and their
 * $/h (a claim launches on the instance type with its cheapest compatible available offering — OrderByPrice's key,
 * types.go:336-355; ties: the lower index). count / cost: n_instance_types doubles each, overwritten.
 * Sharded solves (ksolve_solve_batch over handles on several devices, one process per GPU, ...) SUM these vectors: that sum is
 * the reduction the design calls for — 16 KB at 1000 instance types. Inside one process it is a host loop
 * (ksolve_packing_vector_sum); across processes it is one all-reduce of the same vector (bench.py: torch.distributed, RCCL over
 * xGMI) — latency-bound either way. */
ksolve_status ksolve_packing_vector(const ksolve_handle* h, const ksolve_results* results, double* count, double* cost);
/* The same over n solved problems (handles[i] solved into results[i]; any devices): element-wise sum of their vectors. All
 * handles must share the instance-type catalogue size. */
ksolve_status ksolve_packing_vector_sum(ksolve_handle* const* handles, const ksolve_results* results, uint32_t n, double* count, double* cost);
/* Asks a running ksolve_solve on another thread to stop at the next pod boundary (ctx cancellation). */
ksolve_status ksolve_cancel(ksolve_handle* h);
void ksolve_results_free(ksolve_results* r);
void ksolve_destroy(ksolve_handle* h);
const char* ksolve_last_error(const ksolve_handle* h);
/* ABI version of the loaded library, and whether a gfx950 device is usable (0/1). */
uint32_t ksolve_abi_version(void);
int ksolve_device_available(void);
/* Time of the most recent pack kernel launch measured with HIP events on the handle's stream (milliseconds). */
double ksolve_last_kernel_ms(const ksolve_handle* h, const char* kernel_name);

#ifdef __cplusplus
}
#endif
#endif /* KSOLVE_H */
