//go:build cgo && ksolve

// ksolve_rehydrate.go — ksolve_results -> scheduling.Results (scheduler.go:280-286). The objects handed back are the
// reference's own types, filled to the state the stock Solve() leaves them in after FinalizeScheduling: what the
// provisioner (provisioner.go:430-470), the consolidation simulator (disruption/helpers.go:128-155) and
// Results.Record / TruncateInstanceTypes / AllNonPendingPodsScheduled read.
//
// Lives in package scheduling next to ksolve_flatten.go (see its header). NOT COMPILED IN THIS REPOSITORY'S IMAGE.
package scheduling

/*
#include "ksolve.h"
*/
import "C"

import (
	"fmt"
	"math/big"
	"strconv"
	"unsafe"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/cloudprovider"
	"sigs.k8s.io/karpenter/pkg/scheduling"
)

func cSlice[T any, P any](p *P, n int) []T {
	if p == nil || n == 0 {
		return nil
	}
	return unsafe.Slice((*T)(unsafe.Pointer(p)), n)
}

// quantity: the exact resource.Quantity of a scaled device integer (inverse of quantities.scaled).
func (q *quantities) quantity(dim int, scaled int64) resource.Quantity {
	nano := new(big.Int).Mul(big.NewInt(scaled), q.scale[dim])
	milli, rem := new(big.Int).QuoRem(nano, big.NewInt(1_000_000), new(big.Int))
	if rem.Sign() == 0 && milli.IsInt64() {
		return *resource.NewMilliQuantity(milli.Int64(), lo_format(q.names[dim]))
	}
	out := resource.MustParse(nano.String() + "n")
	return out
}

// lo_format: memory-like resources print in binary SI, everything else in decimal SI — presentation only, Quantity.Cmp
// is what every consumer uses.
func lo_format(name corev1.ResourceName) resource.Format {
	switch name {
	case corev1.ResourceMemory, corev1.ResourceEphemeralStorage:
		return resource.BinarySI
	}
	return resource.DecimalSI
}

// requirements: one row of the claim requirement table back into scheduling.Requirements. Every shape the device can
// produce is one NodeSelectorRequirement operator (plus bounds), which is exactly what NewRequirementWithFlexibility
// and Requirement.Intersection construct on the Go side.
func (f *flatProblem) requirements(cl *C.ksolve_claims, c int) scheduling.Requirements {
	d := f.dict
	nk, rw := int(cl.n_keys), int(cl.req_words)
	mask := cSlice[uint64](cl.req_mask, (c+1)*rw)[c*rw:]
	defined := cSlice[uint32](cl.req_defined, c+1)[c]
	complement := cSlice[uint32](cl.req_complement, c+1)[c]
	hasGte, hasLte := cSlice[uint32](cl.req_has_gte, c+1)[c], cSlice[uint32](cl.req_has_lte, c+1)[c]
	gte, lte := cSlice[int64](cl.req_gte, (c+1)*nk), cSlice[int64](cl.req_lte, (c+1)*nk)
	minValues := cSlice[int32](cl.req_min_values, (c+1)*nk)
	out := scheduling.NewRequirements()
	for k := 0; k < nk; k++ {
		if defined&(1<<uint(k)) == 0 {
			continue
		}
		var values []string
		for i, v := range d.values[k] {
			if mask[int(d.wordOff[k])+i/64]&(1<<uint(i%64)) != 0 {
				values = append(values, v)
			}
		}
		var mv *int
		if minValues != nil && minValues[c*nk+k] >= 0 {
			n := int(minValues[c*nk+k])
			mv = &n
		}
		comp := complement&(1<<uint(k)) != 0
		switch {
		case !comp:
			op := corev1.NodeSelectorOpIn
			if len(values) == 0 {
				op = corev1.NodeSelectorOpDoesNotExist
			}
			out.Add(scheduling.NewRequirementWithFlexibility(d.keys[k], op, mv, values...))
		default:
			// complement sets: NotIn values (Exists when empty), tightened by the bounds — Add() intersects, which is how the
			// reference arrives at the same object (requirement.go:176-225)
			op := corev1.NodeSelectorOpNotIn
			if len(values) == 0 {
				op = corev1.NodeSelectorOpExists
			}
			out.Add(scheduling.NewRequirementWithFlexibility(d.keys[k], op, mv, values...))
			if hasGte&(1<<uint(k)) != 0 {
				out.Add(scheduling.NewRequirementWithFlexibility(d.keys[k], v1.NodeSelectorOpGte, mv, strconv.FormatInt(gte[c*nk+k], 10)))
			}
			if hasLte&(1<<uint(k)) != 0 {
				out.Add(scheduling.NewRequirementWithFlexibility(d.keys[k], v1.NodeSelectorOpLte, mv, strconv.FormatInt(lte[c*nk+k], 10)))
			}
		}
	}
	return out
}

// podError: the error strings of the reference for each device code (include/ksolve.h ksolve_pod_error cites the
// origin of each); callers only branch on IsReservedOfferingError / IsDRAError and print the rest.
func podError(code, diag uint8) error {
	switch C.ksolve_pod_error(code) {
	case C.KSOLVE_POD_TAINTS:
		return fmt.Errorf("did not tolerate taint")
	case C.KSOLVE_POD_INCOMPATIBLE:
		return fmt.Errorf("incompatible requirements")
	case C.KSOLVE_POD_TOPOLOGY:
		return fmt.Errorf("unsatisfiable topology constraint")
	case C.KSOLVE_POD_INSTANCE_TYPES:
		return InstanceTypeFilterError{requirementsMet: diag&1 != 0, fits: diag&2 != 0, hasOffering: diag&4 != 0,
			requirementsAndFits: diag&8 != 0, requirementsAndOffering: diag&16 != 0, fitsAndOffering: diag&32 != 0}
	case C.KSOLVE_POD_RESOURCES:
		return fmt.Errorf("exceeds node resources")
	case C.KSOLVE_POD_NO_TEMPLATES:
		return fmt.Errorf("nodepool requirements filtered out all available instance types")
	case C.KSOLVE_POD_LIMITS:
		return fmt.Errorf("all available instance types exceed limits for nodepool")
	case C.KSOLVE_POD_RESERVED:
		return NewReservedOfferingError(fmt.Errorf("one or more instance types with compatible reserved offerings are available, but could not be reserved"))
	case C.KSOLVE_POD_MIN_VALUES:
		return fmt.Errorf("minValues requirement is not met")
	}
	return fmt.Errorf("pod could not be scheduled")
}

// claimsOf builds the NodeClaims of rows [c0, c1) of a claim table — all of a Solve()'s, or one probe's slice of a sweep's —
// without their pods.
func (f *flatProblem) claimsOf(s *Scheduler, cl *C.ksolve_claims, c0, c1 int) []*NodeClaim {
	nc, nr, iw := int(cl.n_claims), int(cl.n_res), int(cl.it_words)
	tmpl := cSlice[int32](cl.template_idx, nc)
	itMask := cSlice[uint64](cl.it_mask, nc*iw)
	requests := cSlice[int64](cl.requests, nc*nr)
	seq := cSlice[uint32](cl.hostname_seq, nc)
	relaxed := cSlice[uint8](cl.min_values_relaxed, nc)
	reserved := cSlice[uint64](cl.reserved_mask, nc)
	ordered := cSlice[int32](cl.ordered_instance_types, nc*int(cl.n_instance_types))
	orderedCount := cSlice[uint32](cl.ordered_count, nc)

	claims := make([]*NodeClaim, 0, c1-c0)
	for c := c0; c < c1; c++ {
		t := f.templates[tmpl[c]]
		var its []*cloudprovider.InstanceType
		if ordered != nil { // Results.TruncateInstanceTypes already applied on the device (scheduler.go:419-437)
			for _, i := range ordered[c*int(cl.n_instance_types):][:orderedCount[c]] {
				its = append(its, f.its[i])
			}
		} else {
			for i, it := range f.its {
				if itMask[c*iw+i/64]&(1<<uint(i%64)) != 0 {
					its = append(its, it)
				}
			}
		}
		n := NewNodeClaim(t, s.topology, s.daemonOverheadGroups[t], its, s.reservationManager, s.reservedOfferingMode)
		n.hostname = fmt.Sprintf("hostname-placeholder-%04d", seq[c]) // the device's own counter; NewNodeClaim's is process-global
		n.Requirements = f.requirements(cl, c)                        // after FinalizeScheduling: hostname removed, reservation ids injected (nodeclaim.go:353-377)
		reqs := corev1.ResourceList{}
		for r := 0; r < nr; r++ {
			if v := requests[c*nr+r]; v != 0 {
				reqs[f.qty.names[r]] = f.qty.quantity(r, v)
			}
		}
		n.Spec.Resources.Requests = reqs
		if relaxed != nil && relaxed[c] != 0 { // scheduler.go:763-772
			n.Annotations = lo_merge(n.Annotations, map[string]string{v1.NodeClaimMinValuesRelaxedAnnotationKey: "true"})
		}
		if reserved != nil && reserved[c] != 0 {
			kr := int(f.desc.key_reservation_id)
			for _, it := range its {
				for _, o := range it.Offerings {
					if o.CapacityType() == v1.CapacityTypeReserved && o.Available {
						if id, ok := f.dict.valIndex[kr][o.ReservationID()]; ok && reserved[c]&(1<<uint(id)) != 0 {
							n.reservedOfferings = append(n.reservedOfferings, o)
						}
					}
				}
			}
		}
		claims = append(claims, n)
	}

	return claims
}

// rehydrate fills s.newNodeClaims / s.existingNodes exactly where the stock Solve() would have left them and returns
// the Results value that points at them.
func (f *flatProblem) rehydrate(s *Scheduler, res *C.ksolve_results) Results {
	cl := &res.claims
	nc := int(cl.n_claims)
	claims := f.claimsOf(s, cl, 0, nc)

	// pods into their bins, in the order the reference appended them (pod_slot)
	P := int(res.n_pods)
	assign := cSlice[int32](res.pod_assignment, P)
	slot := cSlice[uint32](res.pod_slot, P)
	perr, diag := cSlice[uint8](res.pod_error, P), cSlice[uint8](res.pod_error_diag, P)
	counts := cSlice[uint32](cl.pod_count, nc)
	for c := range claims {
		claims[c].Pods = make([]*corev1.Pod, counts[c])
	}
	nodePods := make([]map[uint32]*corev1.Pod, len(f.nodes))
	podErrors := map[*corev1.Pod]error{}
	for p := 0; p < P; p++ {
		switch a := assign[p]; {
		case a >= 0:
			claims[a].Pods[slot[p]] = f.pods[p]
		case a <= -2:
			e := int(-2 - a)
			if nodePods[e] == nil {
				nodePods[e] = map[uint32]*corev1.Pod{}
			}
			nodePods[e][slot[p]] = f.pods[p]
		default:
			if perr[p] != 0 { // -1 with code 0 = never popped before the deadline: the reference reports no error for it either
				podErrors[f.pods[p]] = podError(perr[p], diag[p])
			}
		}
	}
	for e, m := range nodePods {
		for i := uint32(0); i < uint32(len(m)); i++ {
			f.nodes[e].Pods = append(f.nodes[e].Pods, m[i])
		}
	}
	s.newNodeClaims = claims
	return Results{NewNodeClaims: claims, ExistingNodes: s.existingNodes, PodErrors: podErrors}
}

func lo_merge(a, b map[string]string) map[string]string {
	out := map[string]string{}
	for k, v := range a {
		out[k] = v
	}
	for k, v := range b {
		out[k] = v
	}
	return out
}
