/*
 * ksolve_min.c — the C ABI (include/ksolve.h) used from plain C, without the host flattener: the smallest problem a cgo
 * shim could hand over. Two instance types, one NodePool, five 1.5-cpu pods:
 *
 *   small  2 cpu / 4Gi / 10 pods, 0.10 $/h      big  8 cpu / 16Gi / 10 pods, 0.40 $/h     overhead 100m / 0 / 0
 *
 * The first pod opens a NodeClaim that could still be either type; the second makes it too large for `small`; all five
 * fit `big` (7500m <= 7900m): one NodeClaim, InstanceTypeOptions = {big}.
 *
 *   gcc -std=c99 -Iinclude examples/ksolve_min.c -Lkarpenter_amd -lksolve -o ksolve_min      (needs an MI355X)
 * tests/test_abi.py links the same file against the host emulation of the solver instead (tests/emu, test-only).
 */
#include <stdio.h>
#include <string.h>

#include "ksolve.h"

enum { K_IT = 0, K_ZONE = 1, K_CT = 2, N_KEYS = 3, N_RES = 3, N_ITS = 2, N_PODS = 5 };

int main(void) {
  /* dictionaries: one mask word per key. instance-type: bit 0 small, bit 1 big; zone: bit 0 zone-a; capacity type: bit 0 on-demand */
  const uint32_t key_word_off[N_KEYS + 1] = {0, 1, 2, 3};
  int64_t value_int[3 * 64];
  uint64_t value_is_int[3] = {0, 0, 0};
  memset(value_int, 0, sizeof value_int);

  /* resources in the caller's exact units: cpu in millicores, memory in Mi, pods in pods (SoA: [dimension][entity]) */
  const int64_t it_capacity[N_RES * N_ITS] = {2000, 8000, 4096, 16384, 10, 10};
  const int64_t it_allocatable[N_RES * N_ITS] = {1900, 7900, 4096, 16384, 10, 10};

  /* InstanceType.Requirements: instance-type In [own name], zone In [zone-a], capacity-type In [on-demand] */
  const uint64_t it_mask[N_ITS * 3] = {1, 1, 1, /* small */ 2, 1, 1 /* big */};
  const uint32_t it_defined[N_ITS] = {7, 7}, zeros2[N_ITS] = {0, 0};
  const uint64_t it_avail[N_ITS] = {1, 1}; /* offering cell zone 0 * 4 + capacity type 0 */
  double it_price[N_ITS * 64];
  memset(it_price, 0, sizeof it_price);
  it_price[0 * 64 + 0] = 0.10;
  it_price[1 * 64 + 0] = 0.40;

  /* one NodeClaimTemplate without requirements of its own, offering both types, no taints, no limits */
  const uint64_t tmpl_mask[3] = {0, 0, 0};
  const uint32_t zero1[1] = {0};
  const uint64_t tmpl_taints[1] = {0}, tmpl_its[1] = {3};
  const int64_t tmpl_limits[N_RES + 1] = {0, 0, 0, 0};

  /* five pods, 1500m / 1024Mi / 1 pod each, no requirements, no tolerations, no relaxation variants */
  int64_t pod_requests[N_RES * N_PODS];
  uint64_t pod_mask[N_PODS * 3], pod_tol[N_PODS], uid_hi[N_PODS], uid_lo[N_PODS];
  uint32_t pod_zero[N_PODS];
  int32_t pod_next[N_PODS];
  int64_t pod_creation[N_PODS];
  uint8_t pod_pending[N_PODS], pod_deleting[N_PODS];
  memset(pod_mask, 0, sizeof pod_mask);
  for (int p = 0; p < N_PODS; ++p) {
    pod_requests[0 * N_PODS + p] = 1500; pod_requests[1 * N_PODS + p] = 1024; pod_requests[2 * N_PODS + p] = 1;
    pod_tol[p] = 0; uid_hi[p] = 0; uid_lo[p] = (uint64_t)p + 1; pod_zero[p] = 0; pod_next[p] = -1; pod_creation[p] = 0;
    pod_pending[p] = 1; pod_deleting[p] = 0;
  }

  ksolve_problem_desc d;
  memset(&d, 0, sizeof d);
  d.abi_version = KSOLVE_ABI_VERSION;
  d.n_keys = N_KEYS; d.key_word_off = key_word_off; d.well_known_mask = 7;
  d.key_instance_type = K_IT; d.key_zone = K_ZONE; d.key_capacity_type = K_CT; d.key_hostname = -1;
  d.value_int = value_int; d.value_is_int = value_is_int;
  d.n_res = N_RES;
  d.n_its = N_ITS; d.it_allocatable = it_allocatable; d.it_capacity = it_capacity;
  d.it_reqs.n = N_ITS; d.it_reqs.mask = it_mask; d.it_reqs.defined = it_defined; d.it_reqs.complement = zeros2; d.it_reqs.has_gte = zeros2; d.it_reqs.has_lte = zeros2;
  d.it_offering_avail = it_avail; d.it_offering_price = it_price; d.n_zones = 1; d.n_captypes = 1;
  d.n_templates = 1;
  d.tmpl_reqs.n = 1; d.tmpl_reqs.mask = tmpl_mask; d.tmpl_reqs.defined = zero1; d.tmpl_reqs.complement = zero1; d.tmpl_reqs.has_gte = zero1; d.tmpl_reqs.has_lte = zero1;
  d.tmpl_taints = tmpl_taints; d.tmpl_its = tmpl_its; d.tmpl_limit_mask = zero1; d.tmpl_limits = tmpl_limits;
  d.key_reservation_id = -1; d.captype_reserved = -1;
  d.n_pods = N_PODS; d.n_pod_rows = N_PODS; d.pod_requests = pod_requests;
  d.pod_reqs.n = N_PODS; d.pod_reqs.mask = pod_mask; d.pod_reqs.defined = pod_zero; d.pod_reqs.complement = pod_zero; d.pod_reqs.has_gte = pod_zero; d.pod_reqs.has_lte = pod_zero;
  d.pod_strict_reqs = d.pod_reqs;
  d.pod_tolerates = pod_tol; d.pod_next_variant = pod_next; d.pod_creation = pod_creation; d.pod_uid_hi = uid_hi; d.pod_uid_lo = uid_lo;
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
    printf(" [pods=%u its=0x%llx cpu=%lld price=%.2f]", r.claims.pod_count[c], (unsigned long long)r.claims.it_mask[c * r.claims.it_words],
           (long long)r.claims.requests[c * r.claims.n_res + 0], r.claims.cheapest_price[c]);
  printf(" assignment=");
  for (uint32_t p = 0; p < r.n_pods; ++p) printf("%d", r.pod_assignment[p]);
  printf("\n");
  ksolve_results_free(&r);
  ksolve_destroy(h);
  return 0;
}
