/*
 * ksolve_nodes_topology.c — the C ABI (include/ksolve.h) from plain C for a problem that has the two pieces
 * ksolve_min.c leaves out: an EXISTING NODE (scheduler.go:792-843, existingnode.go:47-75) and a TOPOLOGY GROUP
 * (topologygroup.go:55-77). It is what go/ksolve_flatten.go hands over for:
 *
 *   zones  zone-a, zone-b            instance type  m : 4 cpu / 8Gi / 10 pods, overhead 100m, on-demand 0.20 $/h in both zones
 *   node   node-1 (type m, zone-a, initialized) with 2 cpu / 4Gi / 5 pods still available
 *   pods   four, 1 cpu / 512Mi each, label app=web, topologySpreadConstraints: zone, maxSkew 1, DoNotSchedule, selecting app=web
 *
 * What the reference does (Solve(), scheduler.go:440): pod 1 goes to node-1 (existing nodes first; zone-a has the
 * smallest count, 0). Pod 2 must go to zone-b (skew): node-1 is zone-a, so a NodeClaim is opened and pinned to zone-b.
 * Pod 3: both zones hold one pod, node-1 still has 1 cpu -> node-1. Pod 4: zone-a 2 / zone-b 1 -> the zone-b claim.
 * Result: node-1 gets pods 0 and 2, one NodeClaim {m, zone In [zone-b]} gets pods 1 and 3.
 *
 *   gcc -std=c99 -Iinclude examples/ksolve_nodes_topology.c -Lkarpenter_amd -lksolve -o ksolve_nodes_topology   (needs an MI355X)
 * tests/test_abi.py links it against the host emulation (tests/emu, test-only) and checks the same problem, built by
 * the Python fixtures, against the oracle.
 */
#include <stdio.h>
#include <string.h>

#include "ksolve.h"

enum { K_IT = 0, K_ZONE = 1, K_CT = 2, N_KEYS = 3, N_RES = 3, N_PODS = 4, REQ_WORDS = 3 };

int main(void) {
  /* dictionaries, one mask word per key. instance-type: bit 0 = m; zone: bit 0 = zone-a, bit 1 = zone-b; capacity-type: bit 0 = on-demand */
  const uint32_t key_word_off[N_KEYS + 1] = {0, 1, 2, 3};
  int64_t value_int[REQ_WORDS * 64];
  const uint64_t value_is_int[REQ_WORDS] = {0, 0, 0};
  memset(value_int, 0, sizeof value_int);

  /* the instance type (cpu in millicores, memory in Mi, pods) and its offerings: cell = zone index * 4 + capacity-type index */
  const int64_t it_capacity[N_RES] = {4000, 8192, 10}, it_allocatable[N_RES] = {3900, 8192, 10};
  const uint64_t it_mask[REQ_WORDS] = {1, 3, 1}; /* instance-type In [m], zone In [zone-a, zone-b], capacity-type In [on-demand] */
  const uint32_t seven[1] = {7}, zero1[1] = {0};
  const uint64_t it_avail[1] = {(1ull << 0) | (1ull << 4)};
  double it_price[64];
  memset(it_price, 0, sizeof it_price);
  it_price[0] = 0.20;
  it_price[4] = 0.20;

  /* one NodeClaimTemplate: no requirements of its own, offers m, no taints, no limits */
  const uint64_t tmpl_mask[REQ_WORDS] = {0, 0, 0}, tmpl_taints[1] = {0}, tmpl_its[1] = {1};
  const int64_t tmpl_limits[N_RES + 1] = {0, 0, 0, 0};

  /* the existing node: its labels as single-value requirements (existingnode.go:66), what is still available on it */
  const uint64_t node_mask[REQ_WORDS] = {1, 1, 1}; /* instance-type m, zone-a, on-demand */
  const uint64_t node_taints[1] = {0};
  const int64_t node_remaining[N_RES] = {2000, 4096, 5};
  const uint8_t node_initialized[1] = {1}, node_uca[1] = {0};
  const int32_t node_hostname_value[1] = {-1}; /* no requirement of the problem mentions kubernetes.io/hostname */

  /* the topology group all four pods own and are selected by: spread over the zone key, both zones registered
   * (buildDomainGroups, topology.go:105-146), nothing counted yet, node filter = Honor affinity with the pod's (empty)
   * node selector: one empty requirement set (MakeTopologyNodeFilter, topologynodefilter.go:38-48) */
  const uint8_t tg_type[1] = {0}, tg_inverse[1] = {0}, tg_active[1] = {1}, tg_aff_honor[1] = {1}, tg_taint_honor[1] = {0};
  const int32_t tg_key[1] = {K_ZONE}, tg_skew[1] = {1}, tg_min_domains[1] = {-1};
  const uint64_t tg_domains[1] = {3};
  int32_t tg_counts[64];
  const uint32_t tg_filter_first[2] = {0, 1};
  const uint64_t tg_filter_mask[REQ_WORDS] = {0, 0, 0}, tg_filter_tolerates[1] = {0};
  uint16_t value_rank[REQ_WORDS * 64];
  memset(tg_counts, 0, sizeof tg_counts);
  memset(value_rank, 0, sizeof value_rank);
  value_rank[K_ZONE * 64 + 1] = 1; /* "zone-a" < "zone-b" */

  int64_t pod_requests[N_RES * N_PODS], pod_creation[N_PODS];
  uint64_t pod_mask[N_PODS * REQ_WORDS], pod_tol[N_PODS], uid_hi[N_PODS], uid_lo[N_PODS], pod_owned[N_PODS], pod_selected[N_PODS];
  uint32_t pod_zero[N_PODS];
  int32_t pod_next[N_PODS];
  uint8_t pod_pending[N_PODS], pod_deleting[N_PODS];
  memset(pod_mask, 0, sizeof pod_mask);
  for (int p = 0; p < N_PODS; ++p) {
    pod_requests[0 * N_PODS + p] = 1000; pod_requests[1 * N_PODS + p] = 512; pod_requests[2 * N_PODS + p] = 1;
    pod_tol[p] = 0; uid_hi[p] = 0; uid_lo[p] = (uint64_t)p + 1; pod_zero[p] = 0; pod_next[p] = -1; pod_creation[p] = 0;
    pod_pending[p] = 1; pod_deleting[p] = 0;
    pod_owned[p] = 1; pod_selected[p] = 1; /* bit 0 = the group above */
  }

  ksolve_problem_desc d;
  memset(&d, 0, sizeof d);
  d.abi_version = KSOLVE_ABI_VERSION;
  d.n_keys = N_KEYS; d.key_word_off = key_word_off; d.well_known_mask = 7;
  d.key_instance_type = K_IT; d.key_zone = K_ZONE; d.key_capacity_type = K_CT; d.key_hostname = -1;
  d.value_int = value_int; d.value_is_int = value_is_int;
  d.n_res = N_RES;
  d.n_its = 1; d.it_allocatable = it_allocatable; d.it_capacity = it_capacity;
  d.it_reqs.n = 1; d.it_reqs.mask = it_mask; d.it_reqs.defined = seven; d.it_reqs.complement = zero1; d.it_reqs.has_gte = zero1; d.it_reqs.has_lte = zero1;
  d.it_offering_avail = it_avail; d.it_offering_price = it_price; d.n_zones = 2; d.n_captypes = 1;
  d.n_templates = 1;
  d.tmpl_reqs.n = 1; d.tmpl_reqs.mask = tmpl_mask; d.tmpl_reqs.defined = zero1; d.tmpl_reqs.complement = zero1; d.tmpl_reqs.has_gte = zero1; d.tmpl_reqs.has_lte = zero1;
  d.tmpl_taints = tmpl_taints; d.tmpl_its = tmpl_its; d.tmpl_limit_mask = zero1; d.tmpl_limits = tmpl_limits;
  d.key_reservation_id = -1; d.captype_reserved = -1;

  d.n_nodes = 1;
  d.node_reqs.n = 1; d.node_reqs.mask = node_mask; d.node_reqs.defined = seven; d.node_reqs.complement = zero1; d.node_reqs.has_gte = zero1; d.node_reqs.has_lte = zero1;
  d.node_taints = node_taints; d.node_remaining = node_remaining; d.node_initialized = node_initialized; d.node_under_consolidate_after = node_uca;

  d.topo.n = 1;
  d.topo.type = tg_type; d.topo.inverse = tg_inverse; d.topo.initially_active = tg_active;
  d.topo.key = tg_key; d.topo.max_skew = tg_skew; d.topo.min_domains = tg_min_domains;
  d.topo.domain_words = 1; d.topo.domains = tg_domains; d.topo.init_counts = tg_counts;
  d.topo.filter_affinity_honor = tg_aff_honor; d.topo.filter_taint_honor = tg_taint_honor; d.topo.filter_first = tg_filter_first;
  d.topo.filter_reqs.n = 1; d.topo.filter_reqs.mask = tg_filter_mask; d.topo.filter_reqs.defined = zero1; d.topo.filter_reqs.complement = zero1;
  d.topo.filter_reqs.has_gte = zero1; d.topo.filter_reqs.has_lte = zero1;
  d.topo.filter_tolerates = tg_filter_tolerates; d.topo.value_rank = value_rank; d.topo.node_hostname_value = node_hostname_value;

  d.n_pods = N_PODS; d.n_pod_rows = N_PODS; d.pod_requests = pod_requests;
  d.pod_reqs.n = N_PODS; d.pod_reqs.mask = pod_mask; d.pod_reqs.defined = pod_zero; d.pod_reqs.complement = pod_zero; d.pod_reqs.has_gte = pod_zero; d.pod_reqs.has_lte = pod_zero;
  d.pod_strict_reqs = d.pod_reqs;
  d.pod_tolerates = pod_tol; d.pod_next_variant = pod_next; d.pod_topo_owned = pod_owned; d.pod_topo_selected = pod_selected;
  d.pod_creation = pod_creation; d.pod_uid_hi = uid_hi; d.pod_uid_lo = uid_lo;
  d.pod_is_pending = pod_pending; d.pod_from_deleting_node = pod_deleting;

  ksolve_options o;
  memset(&o, 0, sizeof o);
  o.max_steps = -1;

  ksolve_handle* h = NULL;
  ksolve_status st = ksolve_create(&d, &o, &h);
  if (st != KSOLVE_OK) { fprintf(stderr, "ksolve_create: %d %s\n", (int)st, ksolve_last_error(h)); ksolve_destroy(h); return 1; }
  ksolve_results r;
  st = ksolve_solve(h, &r);
  if (st != KSOLVE_OK) { fprintf(stderr, "ksolve_solve: %d %s\n", (int)st, ksolve_last_error(h)); ksolve_destroy(h); return 1; }
  printf("claims=%u", r.claims.n_claims);
  for (uint32_t c = 0; c < r.claims.n_claims; ++c)
    printf(" [pods=%u its=0x%llx zone=0x%llx cpu=%lld price=%.2f]", r.claims.pod_count[c], (unsigned long long)r.claims.it_mask[c * r.claims.it_words],
           (unsigned long long)r.claims.req_mask[c * r.claims.req_words + key_word_off[K_ZONE]], (long long)r.claims.requests[c * r.claims.n_res + 0],
           r.claims.cheapest_price[c]);
  printf(" assignment=");
  for (uint32_t p = 0; p < r.n_pods; ++p) printf("%s%d", p ? "," : "", r.pod_assignment[p]);
  printf(" errors=");
  for (uint32_t p = 0; p < r.n_pods; ++p) printf("%d", r.pod_error[p]);
  printf("\n");
  ksolve_results_free(&r);
  ksolve_destroy(h);
  return 0;
}
