//go:build cgo && ksolve

// ksolve_sweep.go — a whole consolidation sweep through ksolve_sweep (include/ksolve.h): the cluster is flattened and
// uploaded ONCE per disruption pass, every candidate set is a row of a ksolve_sweep_desc, all simulations run as one launch.
// It replaces n calls of
//
//	disruption.SimulateScheduling(ctx, kubeClient, cluster, provisioner, candidates...)   pkg/controllers/disruption/helpers.go:53-155
//
// made by SingleNodeConsolidation.ComputeCommands (singlenodeconsolidation.go:55-126: one per candidate),
// MultiNodeConsolidation.firstNConsolidationOption (multinodeconsolidation.go:117-207: every prefix candidates[0:k+1] the
// binary search can reach, k = 1..100 — about seven dependent simulations become one launch, the search itself becomes a walk
// over sims[mid]) and the validator (validation.go:297-357). What each caller does with a simulation — computeConsolidation
// (consolidation.go:159-256), filterOutSameInstanceType (multinodeconsolidation.go:209-246), the uninitialized-node rule
// (helpers.go:129-141) — stays the reference's own code: it reads Results, it does not change the simulation.
//
// Added to pkg/controllers/provisioning/scheduling next to ksolve_shim.go. The C++ twin of this file is
// karpenter_amd/host/ksched.cpp (ksched_sweep), which is what this repository's tests and bench.py drive.
//
// Clusters whose pods carry topology constraints sweep the same way (round 4): the resident base counts every bound pod row into
// its groups and hands the domain universe / node registrations apart (ksolve_topology.domain_universe / domain_node_regs, built
// by flattenTopology when flatten() is told which node every row is bound to); a probe then takes its candidates' share out of
// its own copy of the counters on the device (include/ksolve.h, ksolve_probe_create). Every bound pod of a candidate must be
// among `pods`: a pod that a PodDisruptionBudget holds back stays counted under its node in the reference (helpers.go:86-95) —
// keep such a candidate out of the sweep and simulate it with SimulateScheduling.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Go toolchain, SURVEY.md §8c).
package scheduling

/*
#include <stdlib.h>
#include "ksolve.h"
*/
import "C"

import (
	"context"
	"fmt"
	"unsafe"

	corev1 "k8s.io/api/core/v1"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/utils/resources"
)

// ResidentCluster is a cluster that lives on the device for the length of one disruption pass.
type ResidentCluster struct {
	p        *deviceProblem
	nodeOf   map[string]uint32    // StateNode name -> existing-node index of the flat problem
	nodePods map[uint32][]uint32  // pod rows bound to each existing node (the pods a simulation without that node must place)
	always   []uint32             // pod rows of every simulation: pending pods, pods of nodes that are being deleted (helpers.go:69-101)
	limitCap map[uint32][]int64   // per existing node: what its NodePool gets back when the node goes (scheduler.go:835-842), [n_res+1]
	nodeTmpl map[uint32]int       // per existing node: template (NodePool) index, -1 = a pool without limits
	replicas []*ResidentCluster   // the same cluster on other devices (AddReplica): Sweep deals its probes out over all of them
}

// AddReplica registers `other` — NewResidentCluster of the same Scheduler and pods with another ksolve_options.device — as a
// replica: Sweep then runs through ksolve_sweep_replicas, every device simulating a contiguous share of the candidate sets
// (include/ksolve.h). The cluster tables are replicated (0.6 GB at 100k nodes / 2M pods), nothing crosses between devices.
func (rc *ResidentCluster) AddReplica(other *ResidentCluster) { rc.replicas = append(rc.replicas, other) }

// NewResidentCluster flattens `s` — the stock Scheduler assembled over ALL state nodes, no candidate taken out — together with
// every pod some simulation may have to place: pods[i] runs on the node named boundTo[i] ("" = pending, or from a node that is
// being deleted: part of every simulation). The caller builds that list once from the candidates' reschedulable pods
// (helpers.go:63-101 does it per call).
func NewResidentCluster(ctx context.Context, s *Scheduler, pods []*corev1.Pod, boundTo []string) (*ResidentCluster, error) {
	if len(boundTo) != len(pods) {
		return nil, fmt.Errorf("NewResidentCluster: %d pods, %d bindings", len(pods), len(boundTo))
	}
	flat, err := flatten(ctx, s, pods, -1, boundTo)
	if err != nil {
		return nil, err
	}
	rc := &ResidentCluster{nodeOf: map[string]uint32{}, nodePods: map[uint32][]uint32{}, limitCap: map[uint32][]int64{}, nodeTmpl: map[uint32]int{}}
	for e, n := range flat.nodes {
		rc.nodeOf[n.Name()] = uint32(e)
	}
	// pod_node marks the problem as a resident cluster: it is only solved through probes
	podNode := make([]int32, len(pods))
	for i := range pods {
		podNode[i] = -1
		if boundTo[i] == "" {
			rc.always = append(rc.always, uint32(i))
			continue
		}
		e, ok := rc.nodeOf[boundTo[i]]
		if !ok {
			flat.free()
			return nil, fmt.Errorf("pod %s is bound to %s, which is not a state node of the scheduler", pods[i].UID, boundTo[i])
		}
		podNode[i] = int32(e)
		rc.nodePods[e] = append(rc.nodePods[e], uint32(i))
	}
	flat.desc.pod_node = cI32(&flat.arena, podNode)
	// NodePool limits: a removed node's capacity goes back to its pool (scheduler.go:835-842), on the dimensions the pool limits
	nr1 := int(flat.desc.n_res) + 1
	for e, n := range flat.nodes {
		rc.nodeTmpl[uint32(e)] = -1
		for t, tmpl := range flat.templates {
			if tmpl.NodePoolName != n.Labels()[v1.NodePoolLabelKey] {
				continue
			}
			limited := s.remainingResources[tmpl.NodePoolName]
			if len(limited) == 0 {
				break
			}
			back := make([]int64, nr1) // on the dimensions the pool limits, as flatten() encodes tmpl_limits
			for name := range limited {
				if name == resources.Node {
					back[nr1-1] = 1
					continue
				}
				if i, ok := flat.qty.index[name]; ok {
					sv, err := flat.qty.scaled(name, n.Capacity()[name])
					if err != nil {
						flat.free()
						return nil, err
					}
					back[i] = sv
				}
			}
			rc.nodeTmpl[uint32(e)], rc.limitCap[uint32(e)] = t, back
			break
		}
	}
	p := &deviceProblem{flat: flat, s: s}
	if st := C.ksolve_create(&flat.desc, &flat.opts, &p.handle); st != C.KSOLVE_OK {
		msg := "ksolve_create failed"
		if p.handle != nil {
			msg = C.GoString(C.ksolve_last_error(p.handle))
		}
		p.close()
		if st == C.KSOLVE_ERR_UNSUPPORTED {
			return nil, fmt.Errorf("%w: %s", ErrKSolveUnsupported, msg)
		}
		return nil, fmt.Errorf("ksolve_create: %s", msg)
	}
	rc.p = p
	return rc, nil
}

// Close releases the device copy of the cluster.
func (rc *ResidentCluster) Close() { rc.p.close() }

// Simulation is what one SimulateScheduling call returns, for one candidate set of a sweep.
type Simulation struct {
	Results Results
	Err     error // per-simulation status (a probe that needs more NodeClaims than a probe may hold: the caller re-runs that one)
}

// Sweep runs SimulateScheduling for every candidate set (names of the state nodes that are not part of the simulation) in ONE
// ksolve_sweep call and returns the simulations in the order given.
func (rc *ResidentCluster) Sweep(ctx context.Context, candidateSets [][]string) ([]Simulation, error) {
	flat := rc.p.flat
	n := len(candidateSets)
	nodeOff, podOff := make([]uint32, n+1), make([]uint32, n+1)
	var nodes, pods []uint32
	T, nr1 := int(flat.desc.n_templates), int(flat.desc.n_res)+1
	limits := make([]int64, 0, n*T*nr1)
	base := unsafe.Slice((*int64)(unsafe.Pointer(flat.desc.tmpl_limits)), T*nr1)
	for i, set := range candidateSets {
		lim := append([]int64(nil), base...)
		pods = append(pods, rc.always...)
		seen := map[uint32]bool{}
		for _, name := range set {
			e, ok := rc.nodeOf[name]
			if !ok {
				return nil, fmt.Errorf("candidate %s is not a state node of the resident cluster", name)
			}
			if seen[e] { // a candidate named twice is one candidate (the C ABI rejects a node listed twice)
				continue
			}
			seen[e] = true
			nodes = append(nodes, e)
			pods = append(pods, rc.nodePods[e]...)
			if t := rc.nodeTmpl[e]; t >= 0 {
				for r, v := range rc.limitCap[e] {
					lim[t*nr1+r] += v
				}
			}
		}
		nodeOff[i+1], podOff[i+1] = uint32(len(nodes)), uint32(len(pods))
		limits = append(limits, lim...)
	}
	// the descriptors live in C memory for the duration of the call (cgo pointer-passing rules)
	var arena cArena
	defer arena.free()
	var desc C.ksolve_sweep_desc
	desc.n_probes = C.uint32_t(n)
	desc.node_off, desc.nodes = cU32(&arena, nodeOff), cU32(&arena, nodes)
	desc.pod_off, desc.pods = cU32(&arena, podOff), cU32(&arena, pods)
	desc.tmpl_limits = cI64(&arena, limits)
	var out C.ksolve_sweep_results
	// every device's share polls its own handle's flag: a cancelled context reaches all of them (ADVICE r4: only the first
	// device's probes stopped, the call still waited for the slowest replica)
	stops := []func(){watch(ctx, rc.p.handle)}
	for _, r := range rc.replicas {
		stops = append(stops, watch(ctx, r.p.handle))
	}
	stop := func() {
		for _, f := range stops {
			f()
		}
	}
	var st C.ksolve_status
	if len(rc.replicas) == 0 {
		st = C.ksolve_sweep(rc.p.handle, &desc, &out)
	} else {
		handles := (*[1 << 16]*C.ksolve_handle)(arena.bytes((1 + len(rc.replicas)) * int(unsafe.Sizeof(rc.p.handle))))
		handles[0] = rc.p.handle
		for i, r := range rc.replicas {
			handles[1+i] = r.p.handle
		}
		st = C.ksolve_sweep_replicas(&handles[0], C.uint32_t(1+len(rc.replicas)), &desc, &out)
	}
	stop()
	if st != C.KSOLVE_OK && st != C.KSOLVE_ERR_CANCELLED {
		return nil, fmt.Errorf("ksolve_sweep: %s", C.GoString(C.ksolve_last_error(rc.p.handle)))
	}
	defer C.ksolve_sweep_results_free(&out)
	sims := make([]Simulation, n)
	status := unsafe.Slice((*int32)(unsafe.Pointer(out.status)), n)
	claimOff := unsafe.Slice((*uint32)(unsafe.Pointer(out.claim_off)), n+1)
	for i := range sims {
		if status[i] != int32(C.KSOLVE_OK) && status[i] != int32(C.KSOLVE_ERR_CANCELLED) {
			sims[i].Err = fmt.Errorf("ksolve_sweep: simulation %d: status %d", i, status[i])
			continue
		}
		// the probe's slice of the results, re-hydrated exactly like a Solve() of its own (ksolve_rehydrate.go): its claims
		// are rows claimOff[i]..claimOff[i+1] of out.claims, its pods' assignments the slice podOff[i]..podOff[i+1]
		sims[i].Results = flat.rehydrateProbe(rc.p.s, &out, int(claimOff[i]), int(claimOff[i+1]), pods[podOff[i]:podOff[i+1]], int(podOff[i]))
		if status[i] == int32(C.KSOLVE_ERR_CANCELLED) {
			// only a simulation the cancellation actually cut short carries the context's error (Solve returns its partial
			// Results with ctx.Err(), scheduler.go:477-480); the ones that had finished are complete and stay error-free
			if sims[i].Err = ctx.Err(); sims[i].Err == nil {
				sims[i].Err = context.Canceled
			}
		}
	}
	return sims, nil
}

// rehydrateProbe builds scheduling.Results for one probe of a sweep: the same construction as flatProblem.rehydrate, reading
// the probe's claims [c0, c1) of the concatenated claim table and the assignments of its own pods (given by pod row, aligned
// with the descriptor's pod list from position `at`).
func (f *flatProblem) rehydrateProbe(s *Scheduler, out *C.ksolve_sweep_results, c0, c1 int, podRows []uint32, at int) Results {
	assign := unsafe.Slice((*int32)(unsafe.Pointer(out.pod_assignment)), at+len(podRows))[at:]
	perr := unsafe.Slice((*uint8)(unsafe.Pointer(out.pod_error)), at+len(podRows))[at:]
	diag := unsafe.Slice((*uint8)(unsafe.Pointer(out.pod_error_diag)), at+len(podRows))[at:]
	slot := unsafe.Slice((*uint32)(unsafe.Pointer(out.pod_slot)), at+len(podRows))[at:]
	res := Results{PodErrors: map[*corev1.Pod]error{}, NewNodeClaims: f.claimsOf(s, &out.claims, c0, c1)}
	nodePods := map[int][]*corev1.Pod{}
	place := func(dst []*corev1.Pod, at uint32, p *corev1.Pod) []*corev1.Pod { // pod_slot = the position the reference appended the pod at
		for uint32(len(dst)) <= at {
			dst = append(dst, nil)
		}
		dst[at] = p
		return dst
	}
	for j, row := range podRows {
		p := f.pods[row]
		switch a := assign[j]; {
		case a >= 0: // index into THIS probe's claims
			res.NewNodeClaims[a].Pods = place(res.NewNodeClaims[a].Pods, slot[j], p)
		case a <= -2:
			nodePods[int(-2-a)] = place(nodePods[int(-2-a)], slot[j], p)
		default:
			if perr[j] != 0 { // -1 with code 0 = never popped before the deadline: the reference reports no error for it either
				res.PodErrors[p] = podError(perr[j], diag[j])
			}
		}
	}
	// the existing nodes that took pods, as the simulation's own copies (the resident nodes stay pristine for the next sweep);
	// the consumers read Initialized() and Pods of the nodes that received pods (helpers.go:133-153)
	for e, n := range f.nodes {
		if ps := nodePods[e]; len(ps) > 0 {
			en := *n
			en.Pods = ps
			res.ExistingNodes = append(res.ExistingNodes, &en)
		}
	}
	return res
}
