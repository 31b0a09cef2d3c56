//go:build cgo

// Package ksolve is the reference-side binding for the MI355X solver: a drop-in for the two call sites that build and
// run the provisioning scheduler,
//
//	scheduling.NewScheduler(...)   pkg/controllers/provisioning/provisioner.go:359, pkg/controllers/disruption/helpers.go:113
//	(*Scheduler).Solve(ctx, pods)  pkg/controllers/provisioning/provisioner.go:430, pkg/controllers/disruption/helpers.go:128
//
// It keeps the Go types on both sides (cloudprovider.InstanceType in, scheduling.Results out) and moves only the hot
// path behind the C ABI of include/ksolve.h. NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Go toolchain); the C++ host
// library karpenter_amd/host/ksched.cpp performs the identical flattening and is what the tests exercise.
package ksolve

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -L${SRCDIR}/../karpenter_amd -lksolve
#include <stdlib.h>
#include "ksolve.h"
*/
import "C"

import (
	"context"
	"fmt"
	"sort"
	"unsafe"

	corev1 "k8s.io/api/core/v1"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/cloudprovider"
	pscheduling "sigs.k8s.io/karpenter/pkg/controllers/provisioning/scheduling"
	"sigs.k8s.io/karpenter/pkg/scheduling"
)

// Scheduler mirrors scheduling.Scheduler for the accelerated path.
type Scheduler struct {
	handle *C.ksolve_handle
	flat   *flatProblem // C-allocated SoA arrays (no Go pointers cross the boundary) + the dictionaries to rehydrate
	pods   []*corev1.Pod
	its    []*cloudprovider.InstanceType
	pools  []*v1.NodePool
}

// NewScheduler flattens the same inputs scheduling.NewScheduler receives (scheduler.go:127-141) and uploads them.
// It returns (nil, ErrUnsupported) when the problem uses something the device build does not solve; the caller then
// keeps using the stock scheduler for that loop — the device path itself never falls back to a CPU solve.
func NewScheduler(ctx context.Context, nodePools []*v1.NodePool, instanceTypes map[string][]*cloudprovider.InstanceType,
	pods []*corev1.Pod, opts Options) (*Scheduler, error) {
	fp, err := flatten(nodePools, instanceTypes, pods, opts) // dictionary-encode labels, exact int64 resources, PodData, toleration masks, relaxation ladder
	if err != nil {
		return nil, err
	}
	var h *C.ksolve_handle
	st := C.ksolve_create(&fp.desc, &fp.opts, &h)
	if st != C.KSOLVE_OK {
		msg := C.GoString(C.ksolve_last_error(h))
		C.ksolve_destroy(h)
		fp.free()
		if st == C.KSOLVE_ERR_UNSUPPORTED {
			return nil, fmt.Errorf("%w: %s", ErrUnsupported, msg)
		}
		return nil, fmt.Errorf("ksolve_create: %s", msg)
	}
	return &Scheduler{handle: h, flat: fp, pods: pods}, nil
}

// Solve has the signature and error behaviour of (*scheduling.Scheduler).Solve (scheduler.go:440): partial results
// plus ctx.Err() on deadline, per-pod failures in Results.PodErrors.
func (s *Scheduler) Solve(ctx context.Context, pods []*corev1.Pod) (pscheduling.Results, error) {
	done := make(chan struct{})
	go func() { // ctx cancellation -> ksolve_cancel, polled by the pack kernel between pods
		select {
		case <-ctx.Done():
			C.ksolve_cancel(s.handle)
		case <-done:
		}
	}()
	var res C.ksolve_results
	st := C.ksolve_solve(s.handle, &res)
	close(done)
	if st != C.KSOLVE_OK && st != C.KSOLVE_ERR_CANCELLED {
		return pscheduling.Results{}, fmt.Errorf("ksolve_solve: %s", C.GoString(C.ksolve_last_error(s.handle)))
	}
	defer C.ksolve_results_free(&res)
	out := s.rehydrate(&res) // NodeClaims: template copy + Pods in slot order + InstanceTypeOptions from it_mask + Requirements from masks
	return out, ctx.Err()
}

func (s *Scheduler) Close() { C.ksolve_destroy(s.handle); s.flat.free() }

// SolveBatch runs Solve() for several independent Schedulers with ONE launch of the pack kernel (ksolve_solve_batch:
// block b = the wavefront of problem b). This is what disruption.SimulateScheduling's callers want: single-node
// consolidation evaluates one simulation per candidate (singlenodeconsolidation.go:55-126), multi-node consolidation a
// binary search over prefixes (multinodeconsolidation.go:117-207); the candidates' simulations are independent problems.
func SolveBatch(ctx context.Context, scheds []*Scheduler) ([]pscheduling.Results, error) {
	handles := make([]*C.ksolve_handle, len(scheds))
	for i, s := range scheds {
		handles[i] = s.handle
	}
	results := make([]C.ksolve_results, len(scheds))
	st := C.ksolve_solve_batch((**C.ksolve_handle)(unsafe.Pointer(&handles[0])), C.uint32_t(len(scheds)), (*C.ksolve_results)(unsafe.Pointer(&results[0])))
	out := make([]pscheduling.Results, len(scheds))
	for i, s := range scheds {
		if results[i].status == C.KSOLVE_OK || results[i].status == C.KSOLVE_ERR_CANCELLED {
			out[i] = s.rehydrate(&results[i])
		}
		C.ksolve_results_free(&results[i])
	}
	if st != C.KSOLVE_OK {
		return out, fmt.Errorf("ksolve_solve_batch: status %d", int(st))
	}
	return out, ctx.Err()
}

// rehydrate rebuilds scheduling.Results (scheduler.go:281-286) from the flat results.
func (s *Scheduler) rehydrate(res *C.ksolve_results) pscheduling.Results {
	n := int(res.n_pods)
	assign := unsafe.Slice((*int32)(unsafe.Pointer(res.pod_assignment)), n)
	slot := unsafe.Slice((*uint32)(unsafe.Pointer(res.pod_slot)), n)
	code := unsafe.Slice((*uint8)(unsafe.Pointer(res.pod_error)), n)
	members := make([][]int, int(res.claims.n_claims))
	podErrors := map[*corev1.Pod]error{}
	for p := 0; p < n; p++ {
		if a := assign[p]; a >= 0 {
			members[a] = append(members[a], p)
		} else if code[p] != 0 {
			podErrors[s.pods[p]] = podError(code[p], unsafe.Slice((*uint8)(unsafe.Pointer(res.pod_error_diag)), n)[p])
		}
	}
	claims := make([]*pscheduling.NodeClaim, 0, len(members))
	for c, m := range members {
		sort.Slice(m, func(i, j int) bool { return slot[m[i]] < slot[m[j]] })
		claims = append(claims, s.flat.nodeClaim(res, c, m, s.pods)) // InstanceTypeOptions, Requirements (scheduling.Requirements), Spec.Resources.Requests
	}
	// ExistingNodes: assignment <= -2 is existing node (-2 - index) in sortExistingNodes order (scheduler.go:845-858)
	return pscheduling.Results{NewNodeClaims: claims, ExistingNodes: s.flat.existingNodes(res, assign, slot, s.pods), PodErrors: podErrors}
}

var _ = scheduling.NewRequirements // the flattener builds PodData with the reference's own constructors (requirements.go:74-118)
