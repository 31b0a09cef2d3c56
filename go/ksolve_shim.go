//go:build cgo && ksolve

// ksolve_shim.go — the drop-in for the two call sites that build and run the provisioning scheduler:
//
//	scheduling.NewScheduler(...)   pkg/controllers/provisioning/provisioner.go:359, pkg/controllers/disruption/helpers.go:113
//	(*Scheduler).Solve(ctx, pods)  pkg/controllers/provisioning/provisioner.go:430, pkg/controllers/disruption/helpers.go:128
//
// Together with ksolve_flatten.go and ksolve_rehydrate.go it is added to the reference's own package
// pkg/controllers/provisioning/scheduling (build tag `ksolve`); see INTEGRATION.md §2 for the two-line change at each
// call site. It keeps the Go types on both sides and moves only Solve()'s hot path behind include/ksolve.h.
//
// Input assembly is NOT re-implemented: NewDeviceScheduler calls the stock NewScheduler (template prefilter, daemon
// overhead groups, existing nodes, remaining limits, reservation manager are the reference's own code) and the
// flattener reads the state it built. The stock *Scheduler stays inside and is what runs when the device declines a
// problem (KSOLVE_ERR_UNSUPPORTED) — a decision taken here, in the caller; libksolve itself never solves on the CPU.
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
	"errors"
	"fmt"
	"unsafe"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/types"
	"k8s.io/utils/clock"
	"sigs.k8s.io/controller-runtime/pkg/client"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/cloudprovider"
	"sigs.k8s.io/karpenter/pkg/controllers/state"
	"sigs.k8s.io/karpenter/pkg/events"
	"sigs.k8s.io/karpenter/pkg/scheduling"
	"sigs.k8s.io/karpenter/pkg/scheduling/dynamicresources"
)

// DeviceScheduler has Solve() with the signature and error behaviour of (*Scheduler).Solve (scheduler.go:440).
type DeviceScheduler struct {
	stock *Scheduler // assembled by the reference's own NewScheduler; also the path for problems the device declines
}

// NewDeviceScheduler takes exactly NewScheduler's argument list (scheduler.go:127-141).
func NewDeviceScheduler(
	ctx context.Context,
	kubeClient client.Client,
	nodePools []*v1.NodePool,
	cluster *state.Cluster,
	stateNodes []*state.StateNode,
	topology *Topology,
	instanceTypes map[string][]*cloudprovider.InstanceType,
	daemonSetPods []*corev1.Pod,
	recorder events.Recorder,
	clock clock.Clock,
	volumeReqsByPod map[types.UID][]scheduling.Requirements,
	allocator *dynamicresources.Allocator,
	opts ...Options,
) *DeviceScheduler {
	return &DeviceScheduler{stock: NewScheduler(ctx, kubeClient, nodePools, cluster, stateNodes, topology, instanceTypes,
		daemonSetPods, recorder, clock, volumeReqsByPod, allocator, opts...)}
}

// deviceProblem is one uploaded problem: the flat description (C memory), the device handle, and the scheduler it
// was flattened from.
type deviceProblem struct {
	flat   *flatProblem
	handle *C.ksolve_handle
	s      *Scheduler
}

func (p *deviceProblem) close() {
	if p.handle != nil {
		C.ksolve_destroy(p.handle)
	}
	p.flat.free()
}

func upload(ctx context.Context, s *Scheduler, pods []*corev1.Pod) (*deviceProblem, error) {
	flat, err := flatten(ctx, s, pods, -1)
	if err != nil {
		return nil, err
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
	return p, nil
}

// watch turns ctx cancellation into ksolve_cancel (the pack kernel polls the flag between pods, like trySchedule
// polls ctx.Err(), scheduler.go:519). stop() returns only after the goroutine has exited, so ksolve_cancel can never
// run against a handle that the caller has already destroyed.
func watch(ctx context.Context, handles ...*C.ksolve_handle) (stop func()) {
	done, exited := make(chan struct{}), make(chan struct{})
	go func() {
		defer close(exited)
		select {
		case <-ctx.Done():
			for _, h := range handles {
				C.ksolve_cancel(h)
			}
		case <-done:
		}
	}()
	return func() { close(done); <-exited }
}

// Solve: partial results plus ctx.Err() on deadline, per-pod failures in Results.PodErrors — as scheduler.go:440-518.
func (d *DeviceScheduler) Solve(ctx context.Context, pods []*corev1.Pod) (Results, error) {
	p, err := upload(ctx, d.stock, pods)
	if errors.Is(err, ErrKSolveUnsupported) {
		return d.stock.Solve(ctx, pods) // the caller's choice of the stock path; nothing was placed yet, the Scheduler is untouched
	}
	if err != nil {
		return Results{}, err
	}
	defer p.close()
	var res C.ksolve_results
	stop := watch(ctx, p.handle)
	st := C.ksolve_solve(p.handle, &res)
	stop()
	defer C.ksolve_results_free(&res)
	switch st {
	case C.KSOLVE_OK, C.KSOLVE_ERR_CANCELLED:
		return p.flat.rehydrate(d.stock, &res), ctx.Err()
	case C.KSOLVE_ERR_UNSUPPORTED:
		return d.stock.Solve(ctx, pods)
	}
	return Results{}, fmt.Errorf("ksolve_solve: %s", C.GoString(C.ksolve_last_error(p.handle)))
}

// SolveBatch runs Solve() for several independent schedulers with ONE launch of the pack kernel (ksolve_solve_batch:
// block b = the wavefront of problem b). This is what disruption.SimulateScheduling's callers want: single-node
// consolidation evaluates one simulation per candidate (singlenodeconsolidation.go:55-126), multi-node consolidation a
// binary search over prefixes (multinodeconsolidation.go:117-207); the candidates' simulations are independent problems.
// results[i] / errs[i] are what scheds[i].Solve(ctx, pods[i]) would have returned.
func SolveBatch(ctx context.Context, scheds []*DeviceScheduler, pods [][]*corev1.Pod) (results []Results, errs []error) {
	results, errs = make([]Results, len(scheds)), make([]error, len(scheds))
	var onDevice []int
	var problems []*deviceProblem
	for i, d := range scheds {
		p, err := upload(ctx, d.stock, pods[i])
		switch {
		case errors.Is(err, ErrKSolveUnsupported):
			results[i], errs[i] = d.stock.Solve(ctx, pods[i])
		case err != nil:
			errs[i] = err
		default:
			onDevice, problems = append(onDevice, i), append(problems, p)
		}
	}
	if len(problems) == 0 {
		return results, errs
	}
	// handle pointers and result structs live in C memory for the duration of the call (cgo pointer-passing rules)
	n := len(problems)
	handles := (**C.ksolve_handle)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof((*C.ksolve_handle)(nil)))))
	outs := (*C.ksolve_results)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.ksolve_results{}))))
	defer C.free(unsafe.Pointer(handles))
	defer C.free(unsafe.Pointer(outs))
	hs, rs := unsafe.Slice(handles, n), unsafe.Slice(outs, n)
	for j, p := range problems {
		hs[j] = p.handle
	}
	stop := watch(ctx, hs...)
	st := C.ksolve_solve_batch(handles, C.uint32_t(n), outs)
	stop()
	for j, p := range problems {
		i := onDevice[j]
		switch rs[j].status {
		case C.KSOLVE_OK, C.KSOLVE_ERR_CANCELLED:
			results[i], errs[i] = p.flat.rehydrate(p.s, &rs[j]), ctx.Err()
		case C.KSOLVE_ERR_UNSUPPORTED:
			results[i], errs[i] = p.s.Solve(ctx, pods[i])
		default:
			errs[i] = fmt.Errorf("ksolve_solve_batch: problem %d: status %d (batch status %d): %s", i, int(rs[j].status), int(st),
				C.GoString(C.ksolve_last_error(p.handle)))
		}
		C.ksolve_results_free(&rs[j])
		p.close()
	}
	return results, errs
}
