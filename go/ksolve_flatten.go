//go:build cgo && ksolve

// ksolve_flatten.go — the Go twin of karpenter_amd/host/ksched.cpp: turns the state a stock scheduling.Scheduler has
// assembled (NewScheduler, scheduler.go:127-215) plus the pods of one Solve() call into the flat SoA problem of
// include/ksolve.h. It is written to live INSIDE the reference's package
//
//	sigs.k8s.io/karpenter/pkg/controllers/provisioning/scheduling
//
// (copy the three go/ksolve_*.go files there and build with `-tags ksolve`), because the inputs it reads are unexported
// fields of that package: Scheduler.nodeClaimTemplates / existingNodes / remainingResources / daemonOverheadGroups /
// preferences / topology, Topology.topologyGroups / inverseTopologyGroups, TopologyGroup.domains / owners / nodeFilter.
// Nothing here decides a placement: it encodes (dictionaries, exact integer quantities, masks) and nothing else.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Go toolchain, SURVEY.md §8c). The C++ flattener is the tested twin; the
// layout both must produce is pinned by include/ksolve.h and examples/*.c.
package scheduling

/*
#cgo CFLAGS: -I${SRCDIR}/../../../../../ksolve/include
#cgo LDFLAGS: -lksolve
#include <stdlib.h>
#include <string.h>
#include "ksolve.h"
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"math"
	"math/big"
	"sort"
	"strconv"
	"strings"
	"unsafe"

	"github.com/samber/lo"
	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	"k8s.io/apimachinery/pkg/types"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/cloudprovider"
	karpopts "sigs.k8s.io/karpenter/pkg/operator/options"
	"sigs.k8s.io/karpenter/pkg/scheduling"
	"sigs.k8s.io/karpenter/pkg/utils/resources"
)

// ErrKSolveUnsupported: valid Karpenter input that the device build does not solve (ksolve never solves on the CPU
// instead); the caller keeps the stock Scheduler for this loop.
var ErrKSolveUnsupported = errors.New("ksolve: unsupported on the device")

// cArena owns every C allocation of one flat problem: no Go pointer crosses the cgo boundary (cgo pointer-passing rules),
// and everything is released with one free().
type cArena struct{ ptrs []unsafe.Pointer }

func (a *cArena) bytes(n int) unsafe.Pointer {
	if n == 0 {
		n = 8
	}
	p := C.calloc(1, C.size_t(n))
	a.ptrs = append(a.ptrs, p)
	return p
}
func (a *cArena) free() {
	for _, p := range a.ptrs {
		C.free(p)
	}
	a.ptrs = nil
}
func cU64(a *cArena, v []uint64) *C.uint64_t {
	p := a.bytes(8 * len(v))
	copy(unsafe.Slice((*uint64)(p), len(v)), v)
	return (*C.uint64_t)(p)
}
func cI64(a *cArena, v []int64) *C.int64_t {
	p := a.bytes(8 * len(v))
	copy(unsafe.Slice((*int64)(p), len(v)), v)
	return (*C.int64_t)(p)
}
func cU32(a *cArena, v []uint32) *C.uint32_t {
	p := a.bytes(4 * len(v))
	copy(unsafe.Slice((*uint32)(p), len(v)), v)
	return (*C.uint32_t)(p)
}
func cI32(a *cArena, v []int32) *C.int32_t {
	p := a.bytes(4 * len(v))
	copy(unsafe.Slice((*int32)(p), len(v)), v)
	return (*C.int32_t)(p)
}
func cU16(a *cArena, v []uint16) *C.uint16_t {
	p := a.bytes(2 * len(v))
	copy(unsafe.Slice((*uint16)(p), len(v)), v)
	return (*C.uint16_t)(p)
}
func cU8(a *cArena, v []uint8) *C.uint8_t {
	p := a.bytes(len(v))
	copy(unsafe.Slice((*uint8)(p), len(v)), v)
	return (*C.uint8_t)(p)
}
func cF64(a *cArena, v []float64) *C.double {
	p := a.bytes(8 * len(v))
	copy(unsafe.Slice((*float64)(p), len(v)), v)
	return (*C.double)(p)
}

// dictionary: every label key of the problem gets an index, every value of a key a bit in that key's mask words
// (DESIGN.md §3). The instance-type key's dictionary is the instance-type list itself, in catalogue order.
type dictionary struct {
	keys     []string
	keyIndex map[string]int
	values   [][]string
	valIndex []map[string]int
	wordOff  []uint32 // n_keys+1
}

func (d *dictionary) key(k string) int {
	if i, ok := d.keyIndex[k]; ok {
		return i
	}
	d.keyIndex[k] = len(d.keys)
	d.keys = append(d.keys, k)
	d.values = append(d.values, nil)
	d.valIndex = append(d.valIndex, map[string]int{})
	return len(d.keys) - 1
}
func (d *dictionary) value(k int, v string) int {
	if i, ok := d.valIndex[k][v]; ok {
		return i
	}
	d.valIndex[k][v] = len(d.values[k])
	d.values[k] = append(d.values[k], v)
	return len(d.values[k]) - 1
}
func (d *dictionary) observe(r scheduling.Requirements) {
	for _, q := range r.NodeSelectorRequirements() {
		k := d.key(q.Key)
		switch q.Operator {
		case corev1.NodeSelectorOpIn, corev1.NodeSelectorOpNotIn:
			for _, v := range q.Values {
				d.value(k, v)
			}
		}
	}
}
func (d *dictionary) seal() {
	d.wordOff = make([]uint32, len(d.keys)+1)
	for k := range d.keys {
		d.wordOff[k+1] = d.wordOff[k] + uint32((len(d.values[k])+63)/64)
		if len(d.values[k]) == 0 {
			d.wordOff[k+1] = d.wordOff[k] + 1 // a key that only ever appears with Exists / DoesNotExist still owns a word
		}
	}
}
func (d *dictionary) reqWords() int { return int(d.wordOff[len(d.keys)]) }

// reqTable builds a ksolve_reqsets (SoA) from scheduling.Requirements, one entity at a time.
type reqTable struct {
	d                                    *dictionary
	mask                                 []uint64
	defined, complement, hasGte, hasLte  []uint32
	gte, lte                             []int64
	minValues                            []int32
	anyBounds, anyMinValues              bool
}

func (t *reqTable) add(r scheduling.Requirements) {
	d := t.d
	rw, nk := d.reqWords(), len(d.keys)
	base := len(t.mask)
	t.mask = append(t.mask, make([]uint64, rw)...)
	gte, lte, mv := make([]int64, nk), make([]int64, nk), make([]int32, nk)
	for i := range mv {
		mv[i] = -1
	}
	var def, comp, hg, hl uint32
	for _, q := range r.NodeSelectorRequirements() { // In / NotIn / Exists / DoesNotExist / Gte / Lte (Gt and Lt are canonicalised, requirement.go:48-110)
		k := d.keyIndex[q.Key]
		def |= 1 << uint(k)
		if q.MinValues != nil {
			mv[k] = int32(*q.MinValues)
			t.anyMinValues = true
		}
		switch q.Operator {
		case corev1.NodeSelectorOpIn:
			for _, v := range q.Values {
				i := d.valIndex[k][v]
				t.mask[base+int(d.wordOff[k])+i/64] |= 1 << uint(i%64)
			}
		case corev1.NodeSelectorOpNotIn:
			comp |= 1 << uint(k)
			for _, v := range q.Values {
				i := d.valIndex[k][v]
				t.mask[base+int(d.wordOff[k])+i/64] |= 1 << uint(i%64)
			}
		case corev1.NodeSelectorOpExists:
			comp |= 1 << uint(k)
		case corev1.NodeSelectorOpDoesNotExist:
		case v1.NodeSelectorOpGte:
			comp |= 1 << uint(k)
			hg |= 1 << uint(k)
			gte[k], _ = strconv.ParseInt(q.Values[0], 10, 64)
			t.anyBounds = true
		case v1.NodeSelectorOpLte:
			comp |= 1 << uint(k)
			hl |= 1 << uint(k)
			lte[k], _ = strconv.ParseInt(q.Values[0], 10, 64)
			t.anyBounds = true
		}
	}
	t.defined = append(t.defined, def)
	t.complement = append(t.complement, comp)
	t.hasGte = append(t.hasGte, hg)
	t.hasLte = append(t.hasLte, hl)
	t.gte = append(t.gte, gte...)
	t.lte = append(t.lte, lte...)
	t.minValues = append(t.minValues, mv...)
}
func (t *reqTable) c(a *cArena) C.ksolve_reqsets {
	var out C.ksolve_reqsets
	out.n = C.uint32_t(len(t.defined))
	out.mask = cU64(a, t.mask)
	out.defined, out.complement = cU32(a, t.defined), cU32(a, t.complement)
	out.has_gte, out.has_lte = cU32(a, t.hasGte), cU32(a, t.hasLte)
	out.gte, out.lte = cI64(a, t.gte), cI64(a, t.lte)
	out.min_values = cI32(a, t.minValues)
	return out
}

// requirementsIdentity prints a requirement set so that equal sets print alike and different sets differently (every key,
// operator, value and minValues; Requirements.String() hides restricted labels and shortens long value lists).
func requirementsIdentity(r scheduling.Requirements) string {
	reqs := r.NodeSelectorRequirements()
	sort.Slice(reqs, func(i, j int) bool { return reqs[i].Key < reqs[j].Key })
	var b strings.Builder
	for _, q := range reqs {
		vals := append([]string(nil), q.Values...)
		sort.Strings(vals)
		fmt.Fprintf(&b, "%s\x01%s\x01%s\x01", q.Key, q.Operator, strings.Join(vals, "\x02"))
		if q.MinValues != nil {
			fmt.Fprintf(&b, "%d", *q.MinValues)
		}
		b.WriteByte(3)
	}
	return b.String()
}

// quantities: every resource dimension is scaled by the greatest common divisor (in nano units) of all its quantities in
// the problem, so that the device's int64 arithmetic is exact (resource.Quantity is an exact decimal, SURVEY Appendix B3).
type quantities struct {
	names []corev1.ResourceName
	index map[corev1.ResourceName]int
	nano  [][]*big.Int // per dimension, every quantity seen (for the gcd)
	scale []*big.Int
}

func nanoOf(q resource.Quantity) *big.Int {
	dec := q.AsDec() // unscaled * 10^-scale
	n := new(big.Int).Set(dec.UnscaledBig())
	e := 9 - int(dec.Scale())
	ten := big.NewInt(10)
	for ; e > 0; e-- {
		n.Mul(n, ten)
	}
	for ; e < 0; e++ {
		n.Quo(n, ten) // finer than nano never occurs for the quantities Kubernetes validates
	}
	return n
}
func (q *quantities) dim(name corev1.ResourceName) int {
	if i, ok := q.index[name]; ok {
		return i
	}
	q.index[name] = len(q.names)
	q.names = append(q.names, name)
	q.nano = append(q.nano, nil)
	return len(q.names) - 1
}
func (q *quantities) observe(rl corev1.ResourceList) {
	for name, v := range rl {
		i := q.dim(name)
		q.nano[i] = append(q.nano[i], nanoOf(v))
	}
}
func (q *quantities) seal() {
	q.scale = make([]*big.Int, len(q.names))
	for i := range q.names {
		g := new(big.Int)
		for _, n := range q.nano[i] {
			g.GCD(nil, nil, g, new(big.Int).Abs(n))
		}
		if g.Sign() == 0 {
			g.SetInt64(1)
		}
		q.scale[i] = g
	}
}
func (q *quantities) scaled(name corev1.ResourceName, v resource.Quantity) (int64, error) {
	i := q.index[name]
	n := new(big.Int).Quo(nanoOf(v), q.scale[i])
	if !n.IsInt64() {
		return 0, fmt.Errorf("%w: quantity %s of %s does not fit 63 bits after scaling", ErrKSolveUnsupported, v.String(), name)
	}
	return n.Int64(), nil
}
func (q *quantities) vector(rl corev1.ResourceList, out []int64, stride, at int) error {
	for name, v := range rl {
		i, ok := q.index[name]
		if !ok {
			continue
		}
		s, err := q.scaled(name, v)
		if err != nil {
			return err
		}
		out[i*stride+at] = s
	}
	return nil
}

// podRow is one row of the pod table: the pod as submitted, or one step of its relaxation ladder (preferences.go:38-57).
type podRow struct {
	pod  *corev1.Pod
	data *PodData
	next int32
}

// flatProblem: the C description, what is needed to rehydrate Results, and the arena that owns the C memory.
type flatProblem struct {
	arena     cArena
	desc      C.ksolve_problem_desc
	opts      C.ksolve_options
	dict      *dictionary
	qty       *quantities
	pods      []*corev1.Pod
	its       []*cloudprovider.InstanceType // catalogue order = instance-type key dictionary
	templates []*NodeClaimTemplate
	nodes     []*ExistingNode
	// resident clusters (ksolve_sweep.go): the StateNode name every pod row is bound to ("" = pending); nil otherwise
	boundTo []string
}

func (f *flatProblem) free() { f.arena.free() }

// flatten encodes scheduler s (already assembled by NewScheduler) and the pods of this Solve().
// nolint:gocyclo
func flatten(ctx context.Context, s *Scheduler, pods []*corev1.Pod, maxSteps int64, boundTo ...[]string) (*flatProblem, error) {
	if s.allocator != nil {
		return nil, fmt.Errorf("%w: dynamic resource allocation", ErrKSolveUnsupported)
	}
	f := &flatProblem{dict: &dictionary{keyIndex: map[string]int{}}, qty: &quantities{index: map[corev1.ResourceName]int{}}, pods: pods,
		templates: s.nodeClaimTemplates, nodes: s.existingNodes}
	if len(boundTo) == 1 {
		f.boundTo = boundTo[0] // a resident cluster: the topology tables then count the bound pod rows too (flattenTopology)
	}
	d, q := f.dict, f.qty
	a := &f.arena
	q.dim(corev1.ResourceCPU)    // dimension 0 and 1 are what the queue sorts on (queue.go:72-90)
	q.dim(corev1.ResourceMemory)
	q.dim(corev1.ResourcePods)

	// ---- the catalogue: the union of the NodePools' instance types, first-seen order; key_it's dictionary is this list ----
	keyIT := d.key(corev1.LabelInstanceTypeStable)
	itIndex := map[string]int{}
	for _, t := range s.nodeClaimTemplates {
		for _, it := range s.instanceTypes[t.NodePoolName] {
			if _, ok := itIndex[it.Name]; ok {
				continue
			}
			itIndex[it.Name] = len(f.its)
			f.its = append(f.its, it)
			d.value(keyIT, it.Name)
		}
	}
	keyZone, keyCT := d.key(corev1.LabelTopologyZone), d.key(v1.CapacityTypeLabelKey)

	// ---- pod rows: PodData of every pod and of every step of its relaxation ladder (scheduler.go:521-552) ----
	rows := make([]podRow, 0, len(pods))
	for _, p := range pods {
		s.updateCachedPodData(ctx, p)
		rows = append(rows, podRow{pod: p, data: s.cachedPodData[p.UID], next: -1})
	}
	for i := range pods {
		cur := i
		relaxed := pods[i].DeepCopy()
		for s.preferences.Relax(ctx, relaxed) {
			s.updateCachedPodData(ctx, relaxed)
			rows[cur].next = int32(len(rows))
			rows = append(rows, podRow{pod: relaxed.DeepCopy(), data: s.cachedPodData[relaxed.UID], next: -1})
			cur = len(rows) - 1
		}
		s.updateCachedPodData(ctx, pods[i]) // the cache must describe the pod as submitted again
	}

	// ---- pass 1: dictionaries and quantity scales over everything the problem mentions ----
	for _, it := range f.its {
		d.observe(it.Requirements)
		for _, o := range it.Offerings {
			d.observe(o.Requirements)
			if (len(o.CapacityOverride) > 0 || o.OverheadOverride != nil) && o.CapacityType() == v1.CapacityTypeReserved {
				return nil, fmt.Errorf("%w: capacity / overhead overrides on a reserved offering", ErrKSolveUnsupported)
			}
		}
		q.observe(it.Capacity)
		for _, g := range it.AllocatableOfferingsList() { // base group first, then one group per distinct override pair (types.go:222-269)
			q.observe(g.Allocatable)
		}
	}
	for _, t := range s.nodeClaimTemplates {
		d.observe(t.Requirements)
		q.observe(s.remainingResources[t.NodePoolName])
		for _, g := range s.daemonOverheadGroups[t] {
			q.observe(g.DaemonOverhead)
		}
	}
	for _, r := range rows {
		if r.data.HasResourceClaimRequests {
			return nil, fmt.Errorf("%w: pod %s/%s has resource claims", ErrKSolveUnsupported, r.pod.Namespace, r.pod.Name)
		}
		for _, alt := range r.data.VolumeRequirements { // volumeReqsByPod[uid] (scheduler.go:572): the alternatives, in order
			d.observe(alt)
		}
		d.observe(r.data.Requirements)
		d.observe(r.data.StrictRequirements)
		q.observe(r.data.Requests)
	}
	for _, n := range s.existingNodes {
		d.observe(n.requirements)
		q.observe(n.remainingResources)
	}
	groups, inverse := orderedTopologyGroups(s.topology)
	for _, g := range append(append([]*TopologyGroup{}, groups...), inverse...) {
		if g.Key != corev1.LabelHostname {
			k := d.key(g.Key)
			for dom := range g.domains {
				d.value(k, dom)
			}
		}
		for _, r := range g.nodeFilter.Requirements {
			d.observe(r)
		}
	}
	if len(d.keys) > int(C.KSOLVE_MAX_KEYS) || len(q.names) > int(C.KSOLVE_MAX_RES) || len(f.its) > 64*int(C.KSOLVE_MAX_ITWORDS) {
		return nil, fmt.Errorf("%w: %d keys / %d resources / %d instance types", ErrKSolveUnsupported, len(d.keys), len(q.names), len(f.its))
	}
	d.seal()
	q.seal()
	nk, rw, nr, nIT := len(d.keys), d.reqWords(), len(q.names), len(f.its)
	itWords := (nIT + 63) / 64

	desc := &f.desc
	desc.abi_version = C.KSOLVE_ABI_VERSION
	desc.n_keys = C.uint32_t(nk)
	desc.key_word_off = cU32(a, d.wordOff)
	var wk uint32
	for k, name := range d.keys {
		if v1.WellKnownLabels.Has(name) {
			wk |= 1 << uint(k)
		}
	}
	desc.well_known_mask = C.uint32_t(wk)
	desc.key_instance_type, desc.key_zone, desc.key_capacity_type = C.int32_t(keyIT), C.int32_t(keyZone), C.int32_t(keyCT)
	desc.key_hostname = -1
	if k, ok := d.keyIndex[corev1.LabelHostname]; ok {
		desc.key_hostname = C.int32_t(k)
	}
	valueInt, valueIsInt := make([]int64, rw*64), make([]uint64, rw)
	for k := range d.keys {
		for i, v := range d.values[k] {
			if n, err := strconv.Atoi(v); err == nil { // requirement.go:339
				valueInt[(int(d.wordOff[k])*64)+i] = int64(n)
				valueIsInt[int(d.wordOff[k])+i/64] |= 1 << uint(i%64)
			}
		}
	}
	desc.value_int, desc.value_is_int = cI64(a, valueInt), cU64(a, valueIsInt)
	desc.n_res = C.uint32_t(nr)

	// ---- instance types ----
	nz, nct := len(d.values[keyZone]), len(d.values[keyCT])
	if nz > int(C.KSOLVE_MAX_ZONES) || nct > int(C.KSOLVE_MAX_CAPTYPES) {
		return nil, fmt.Errorf("%w: %d zones / %d capacity types", ErrKSolveUnsupported, nz, nct)
	}
	alloc, capacity := make([]int64, nr*nIT), make([]int64, nr*nIT)
	avail, price := make([]uint64, nIT), make([]float64, nIT*64)
	baseAvail := make([]uint64, nIT)
	var xgIT []uint32
	var xgAvail []uint64
	var xgAlloc [][]int64 // per extra group: nr values
	cellOf := func(o *cloudprovider.Offering) int { return d.valIndex[keyZone][o.Zone()]*4 + d.valIndex[keyCT][o.CapacityType()] }
	itReqs := &reqTable{d: d}
	for i, it := range f.its {
		if err := q.vector(it.Allocatable(), alloc, nIT, i); err != nil {
			return nil, err
		}
		if err := q.vector(it.Capacity, capacity, nIT, i); err != nil {
			return nil, err
		}
		itReqs.add(it.Requirements)
		for _, o := range it.Offerings { // reserved offerings sit in their cell too and, in detail, in their own CSR (below)
			if !o.Available {
				continue
			}
			cell := cellOf(o)
			if avail[i]&(1<<uint(cell)) == 0 || o.Price < price[i*64+cell] {
				price[i*64+cell] = o.Price
			}
			avail[i] |= 1 << uint(cell)
		}
		// the reference has already grouped the available offerings by override (AllocatableOfferingsList): group 0 is the
		// base group (it_allocatable above), every other group becomes one override row
		for g, group := range it.AllocatableOfferingsList() {
			var cells uint64
			for _, o := range group.Offerings {
				cells |= 1 << uint(cellOf(o))
			}
			if g == 0 {
				baseAvail[i] = cells
				continue
			}
			row := make([]int64, nr)
			if err := q.vector(group.Allocatable, row, 1, 0); err != nil {
				return nil, err
			}
			xgIT, xgAvail, xgAlloc = append(xgIT, uint32(i)), append(xgAvail, cells), append(xgAlloc, row)
		}
	}
	// reserved offerings: a CSR per instance type; the reservation index is the id's value index in ReservationIDLabel's
	// dictionary, its capacity the smallest over the id's offerings (NewReservationManager, reservationmanager.go:33-55)
	desc.key_reservation_id, desc.captype_reserved = -1, -1
	if kr, ok := d.keyIndex[cloudprovider.ReservationIDLabel]; ok && len(d.values[kr]) > 0 {
		if len(d.values[kr]) > 64 {
			return nil, fmt.Errorf("%w: %d capacity reservations", ErrKSolveUnsupported, len(d.values[kr]))
		}
		resCap := make([]int32, len(d.values[kr]))
		seen := make([]bool, len(resCap))
		first := make([]uint32, nIT+1)
		var rZone, rID []uint8
		var rPrice []float64
		for i, it := range f.its {
			first[i] = uint32(len(rID))
			for _, o := range it.Offerings {
				if o.CapacityType() != v1.CapacityTypeReserved {
					continue
				}
				id := d.valIndex[kr][o.ReservationID()]
				if !seen[id] || resCap[id] > int32(o.ReservationCapacity) {
					resCap[id], seen[id] = int32(o.ReservationCapacity), true
				}
				if !o.Available {
					continue
				}
				rZone = append(rZone, uint8(d.valIndex[keyZone][o.Zone()]))
				rID = append(rID, uint8(id))
				rPrice = append(rPrice, o.Price)
			}
		}
		first[nIT] = uint32(len(rID))
		desc.n_reservations, desc.reservation_capacity = C.uint32_t(len(resCap)), cI32(a, resCap)
		desc.key_reservation_id = C.int32_t(kr)
		if ct, ok := d.valIndex[keyCT][v1.CapacityTypeReserved]; ok {
			desc.captype_reserved = C.int32_t(ct)
		}
		desc.it_reserved_first = cU32(a, first)
		desc.reserved_zone, desc.reserved_id, desc.reserved_price = cU8(a, rZone), cU8(a, rID), cF64(a, rPrice)
	}
	desc.n_its = C.uint32_t(nIT)
	desc.it_allocatable, desc.it_capacity = cI64(a, alloc), cI64(a, capacity)
	desc.it_reqs = itReqs.c(a)
	desc.it_offering_avail, desc.it_offering_price = cU64(a, avail), cF64(a, price)
	if nx := len(xgIT); nx > 0 {
		if nx > int(C.KSOLVE_MAX_OVERRIDE_GROUPS) {
			return nil, fmt.Errorf("%w: %d offering override groups", ErrKSolveUnsupported, nx)
		}
		soa := make([]int64, nr*nx)
		for e, row := range xgAlloc {
			for r, v := range row {
				soa[r*nx+e] = v
			}
		}
		desc.n_override_groups = C.uint32_t(nx)
		desc.override_it, desc.override_allocatable, desc.override_avail = cU32(a, xgIT), cI64(a, soa), cU64(a, xgAvail)
		desc.it_base_avail = cU64(a, baseAvail)
	}
	desc.n_zones, desc.n_captypes = C.uint32_t(nz), C.uint32_t(nct)

	// ---- distinct taints: a pod's toleration mask and a template's / node's taint mask are bits over this list ----
	var taints []corev1.Taint
	taintBit := func(t corev1.Taint) int {
		for i := range taints {
			if taints[i].MatchTaint(&t) && taints[i].Value == t.Value {
				return i
			}
		}
		taints = append(taints, t)
		return len(taints) - 1
	}
	taintMask := func(ts []corev1.Taint) uint64 {
		var m uint64
		for _, t := range ts {
			if t.Effect == corev1.TaintEffectPreferNoSchedule { // never blocks scheduling (taints.go:83-95 checks NoSchedule / NoExecute)
				continue
			}
			m |= 1 << uint(taintBit(t))
		}
		return m
	}

	// ---- templates (already in OrderByWeight order, already prefiltered: scheduler.go:156-171) ----
	T := len(s.nodeClaimTemplates)
	if T > int(C.KSOLVE_MAX_TEMPLATES) {
		return nil, fmt.Errorf("%w: %d NodePools", ErrKSolveUnsupported, T)
	}
	tmplReqs := &reqTable{d: d}
	tmplTaints, tmplIts := make([]uint64, T), make([]uint64, T*itWords)
	limitMask, limits := make([]uint32, T), make([]int64, T*(nr+1))
	var dgFirst []uint32
	var dgIts []uint64
	var dgOverhead []int64
	var dgNonEmpty []uint8
	anyDaemons := false
	for ti, t := range s.nodeClaimTemplates {
		tmplReqs.add(t.Requirements)
		tmplTaints[ti] = taintMask(t.Spec.Taints)
		for _, it := range s.instanceTypes[t.NodePoolName] { // the pool's full list: the device repeats NewScheduler's prefilter itself
			i := itIndex[it.Name]
			tmplIts[ti*itWords+i/64] |= 1 << uint(i%64)
		}
		for name, v := range s.remainingResources[t.NodePoolName] {
			if name == resources.Node {
				limitMask[ti] |= 1 << uint(nr)
				limits[ti*(nr+1)+nr] = v.Value()
				continue
			}
			if i, ok := q.index[name]; ok {
				sv, err := q.scaled(name, v)
				if err != nil {
					return nil, err
				}
				limitMask[ti] |= 1 << uint(i)
				limits[ti*(nr+1)+i] = sv
			}
		}
		dgFirst = append(dgFirst, uint32(len(dgNonEmpty)))
		for _, g := range s.daemonOverheadGroups[t] {
			mask := make([]uint64, itWords)
			for _, it := range g.InstanceTypes {
				i := itIndex[it.Name]
				mask[i/64] |= 1 << uint(i%64)
			}
			dgIts = append(dgIts, mask...)
			ov := make([]int64, nr)
			if err := q.vector(g.DaemonOverhead, ov, 1, 0); err != nil {
				return nil, err
			}
			dgOverhead = append(dgOverhead, ov...)
			dgNonEmpty = append(dgNonEmpty, lo.Ternary(len(g.DaemonOverhead) > 0, uint8(1), uint8(0)))
			anyDaemons = anyDaemons || len(g.DaemonOverhead) > 0
		}
	}
	dgFirst = append(dgFirst, uint32(len(dgNonEmpty)))
	desc.n_templates = C.uint32_t(T)
	desc.tmpl_reqs = tmplReqs.c(a)
	desc.tmpl_its = cU64(a, tmplIts)
	desc.tmpl_limit_mask, desc.tmpl_limits = cU32(a, limitMask), cI64(a, limits)
	if anyDaemons {
		if len(dgNonEmpty) > 64 {
			return nil, fmt.Errorf("%w: %d daemon-overhead groups", ErrKSolveUnsupported, len(dgNonEmpty))
		}
		desc.tmpl_daemon_first, desc.daemon_group_its = cU32(a, dgFirst), cU64(a, dgIts)
		desc.daemon_group_overhead, desc.daemon_group_nonempty = cI64(a, dgOverhead), cU8(a, dgNonEmpty)
	}

	// ---- existing nodes, in sortExistingNodes order (scheduler.go:845-858) ----
	E := len(s.existingNodes)
	nodeReqs := &reqTable{d: d}
	nodeTaints, nodeRemaining := make([]uint64, E), make([]int64, nr*E)
	nodeInit, nodeUCA := make([]uint8, E), make([]uint8, E)
	for e, n := range s.existingNodes {
		nodeReqs.add(n.requirements)
		nodeTaints[e] = taintMask(n.cachedTaints)
		if err := q.vector(n.remainingResources, nodeRemaining, E, e); err != nil {
			return nil, err
		}
		nodeInit[e] = lo.Ternary(n.Initialized(), uint8(1), uint8(0))
		nodeUCA[e] = lo.Ternary(n.isUnderConsolidateAfter, uint8(1), uint8(0))
	}
	desc.n_nodes = C.uint32_t(E)
	desc.node_reqs = nodeReqs.c(a)
	desc.node_taints, desc.node_remaining = cU64(a, nodeTaints), cI64(a, nodeRemaining)
	desc.node_initialized, desc.node_under_consolidate_after = cU8(a, nodeInit), cU8(a, nodeUCA)

	// ---- pods ----
	P, R := len(pods), len(rows)
	requests := make([]int64, nr*R)
	podReqs, strictReqs := &reqTable{d: d}, &reqTable{d: d}
	tolerates, next := make([]uint64, R), make([]int32, R)
	for r, row := range rows {
		if err := q.vector(row.data.Requests, requests, R, r); err != nil {
			return nil, err
		}
		podReqs.add(row.data.Requirements)
		strictReqs.add(row.data.StrictRequirements)
		next[r] = row.next
	}
	// the distinct taints are only known once templates and nodes have been seen: toleration masks last (taints.go:83-95)
	if len(taints) > 64 {
		return nil, fmt.Errorf("%w: %d distinct taints", ErrKSolveUnsupported, len(taints))
	}
	for r, row := range rows {
		for i := range taints {
			if scheduling.Taints([]corev1.Taint{taints[i]}).ToleratesPod(row.pod) == nil {
				tolerates[r] |= 1 << uint(i)
			}
		}
	}
	// ---- host ports (hostportusage.go:39-117): bits over the distinct <hostIP, hostPort, protocol> triples; only when a
	// pod of this Solve binds one. A row's conflict mask is every triple of the dictionary that Matches one of its own.
	var hpDict []scheduling.HostPort
	hpMask := func(ports []scheduling.HostPort) (uint64, error) {
		var m uint64
		for _, p := range ports {
			i := 0
			for ; i < len(hpDict); i++ {
				if hpDict[i].IP.Equal(p.IP) && hpDict[i].Port == p.Port && hpDict[i].Protocol == p.Protocol {
					break
				}
			}
			if i == len(hpDict) {
				if i == 64 {
					return 0, fmt.Errorf("%w: more than 64 distinct host ports", ErrKSolveUnsupported)
				}
				hpDict = append(hpDict, p)
			}
			m |= 1 << uint(i)
		}
		return m, nil
	}
	hpOn := false
	for _, row := range rows {
		hpOn = hpOn || len(scheduling.GetHostPorts(row.pod)) > 0
	}
	if hpOn {
		use := make([]uint64, R)
		var err error
		for r, row := range rows {
			if use[r], err = hpMask(scheduling.GetHostPorts(row.pod)); err != nil {
				return nil, err
			}
		}
		nodeHP := make([]uint64, max(E, 1))
		for e, n := range s.existingNodes {
			if nodeHP[e], err = hpMask(n.HostPortUsage().Reserved()); err != nil { // accessor added by go/hostportusage_ksolve.go
				return nil, err
			}
		}
		var groupHP []uint64
		for _, t := range s.nodeClaimTemplates {
			for _, g := range s.daemonOverheadGroups[t] {
				m, err := hpMask(g.HostPortUsage.Reserved())
				if err != nil {
					return nil, err
				}
				groupHP = append(groupHP, m)
			}
		}
		conf := make([]uint64, R)
		for r := range rows {
			for i := range hpDict {
				if use[r]&(1<<uint(i)) == 0 {
					continue
				}
				for j := range hpDict {
					if hpDict[i].Matches(hpDict[j]) {
						conf[r] |= 1 << uint(j)
					}
				}
			}
		}
		desc.pod_host_ports, desc.pod_host_port_conflicts = cU64(a, use), cU64(a, conf)
		if E > 0 {
			desc.node_host_ports = cU64(a, nodeHP)
		}
		if anyDaemons {
			desc.daemon_group_host_ports = cU64(a, groupHP)
		}
	}
	// ---- CSI volume limits of existing nodes (VolumeUsage.ExceedsLimits / Add, pkg/scheduling/volumeusage.go:193-209; checked by
	// ExistingNode.CanAdd at existingnode.go:88, updated by Add at :179). A volume is a distinct <driver, PVC> pair; only drivers
	// with a limit on some node are tracked. Same tables as the C++ twin (karpenter_amd/host/ksched.cpp).
	if E > 0 {
		driverIdx := map[string]int{}
		for _, n := range s.existingNodes {
			_, limits := n.VolumeUsage().Tracked() // accessor added by go/volumeusage_ksolve.go
			for drv := range limits {
				if _, ok := driverIdx[drv]; !ok {
					driverIdx[drv] = len(driverIdx)
				}
			}
		}
		if len(driverIdx) > C.KSOLVE_MAX_VOLUME_DRIVERS {
			return nil, fmt.Errorf("%w: %d CSI drivers with volume limits", ErrKSolveUnsupported, len(driverIdx))
		}
		if nd := len(driverIdx); nd > 0 {
			type volKey struct {
				drv int
				pvc string
			}
			volID := map[volKey]uint32{}
			var volDriver []uint8
			vid := func(drv, pvc string) (uint32, bool) {
				di, ok := driverIdx[drv]
				if !ok {
					return 0, false
				}
				k := volKey{di, pvc}
				if id, ok := volID[k]; ok {
					return id, true
				}
				id := uint32(len(volID))
				volID[k] = id
				volDriver = append(volDriver, uint8(di))
				return id, true
			}
			podFirst := make([]uint32, len(pods)+1)
			var podPVs []uint32
			for i, p := range pods { // rows [0, len(pods)) are the pods as submitted; variant rows share their pod's volumes
				podFirst[i] = uint32(len(podPVs))
				vols, err := scheduling.GetVolumes(ctx, s.kubeClient, p) // scheduler.go:622-626
				if err != nil {
					return nil, err
				}
				seen := map[uint32]bool{}
				for drv, pvcs := range vols {
					for pvc := range pvcs {
						if id, ok := vid(drv, pvc); ok && !seen[id] {
							seen[id] = true
							podPVs = append(podPVs, id)
						}
					}
				}
				if len(seen) > 64 {
					return nil, fmt.Errorf("%w: pod %s/%s mounts more than 64 volumes under CSI limits", ErrKSolveUnsupported, p.Namespace, p.Name)
				}
			}
			podFirst[len(pods)] = uint32(len(podPVs))
			nodeFirst := make([]uint32, E+1)
			overLimit := false
			var nodePVs []uint32
			nodeLimit := make([]int32, E*nd)
			for i := range nodeLimit {
				nodeLimit[i] = -1
			}
			for e, n := range s.existingNodes {
				nodeFirst[e] = uint32(len(nodePVs))
				volumes, limits := n.VolumeUsage().Tracked()
				var ids []uint32
				used := make([]int, nd)
				for drv, pvcs := range volumes {
					for pvc := range pvcs {
						if id, ok := vid(drv, pvc); ok {
							ids = append(ids, id)
							used[driverIdx[drv]]++
						}
					}
				}
				sort.Slice(ids, func(i, j int) bool { return ids[i] < ids[j] })
				nodePVs = append(nodePVs, ids...)
				for drv, lim := range limits {
					nodeLimit[e*nd+driverIdx[drv]] = int32(lim)
					if used[driverIdx[drv]] > lim {
						// over a limit already: ExceedsLimits rejects every pod (it walks the drivers of the union) — the
						// device sees a node without room
						nodeRemaining[0*E+e] = -1
						overLimit = true
					}
				}
			}
			if overLimit {
				desc.node_remaining = cI64(a, nodeRemaining) // the table was copied before this block
			}
			nodeFirst[E] = uint32(len(nodePVs))
			if len(podPVs) == 0 {
				podPVs = []uint32{0}
			}
			if len(nodePVs) == 0 {
				nodePVs = []uint32{0}
			}
			desc.n_volume_drivers, desc.n_volumes = C.uint32_t(nd), C.uint32_t(len(volDriver))
			desc.volume_driver = cU8(a, volDriver)
			desc.pod_pv_first, desc.pod_pvs = cU32(a, podFirst), cU32(a, podPVs)
			desc.node_pv_first, desc.node_pvs = cU32(a, nodeFirst), cU32(a, nodePVs)
			desc.node_pv_limit = cI32(a, nodeLimit)
		}
	}
	// ---- volume requirement alternatives (PodData.VolumeRequirements; nodeclaim.go:138-157, existingnode.go:108-139): one
	// requirement set per alternative, list after list; pods with equal lists share (first, count) — it is part of a pod's
	// class identity on the device. A relaxed row carries its pod's list (updateCachedPodData copies it by UID).
	anyVolumes := false
	for _, row := range rows {
		anyVolumes = anyVolumes || len(row.data.VolumeRequirements) > 0
	}
	if anyVolumes {
		volReqs := &reqTable{d: d}
		lists := map[string][2]uint32{}
		first, count := make([]uint32, R), make([]uint32, R)
		for r, row := range rows {
			alts := row.data.VolumeRequirements
			if len(alts) == 0 {
				continue
			}
			var key strings.Builder
			for _, alt := range alts {
				key.WriteString(requirementsIdentity(alt))
				key.WriteByte(0)
			}
			fc, ok := lists[key.String()]
			if !ok {
				fc = [2]uint32{uint32(len(volReqs.defined)), uint32(len(alts))}
				for _, alt := range alts {
					volReqs.add(alt)
				}
				lists[key.String()] = fc
			}
			first[r], count[r] = fc[0], fc[1]
		}
		if volReqs.anyMinValues {
			return nil, fmt.Errorf("%w: volume requirements with minValues", ErrKSolveUnsupported)
		}
		desc.n_volume_reqs = C.uint32_t(len(volReqs.defined))
		desc.volume_reqs = volReqs.c(a)
		desc.pod_volume_first, desc.pod_volume_count = cU32(a, first), cU32(a, count)
	}
	desc.tmpl_taints = cU64(a, tmplTaints)
	desc.n_taints = C.uint32_t(len(taints))
	creation, uidHi, uidLo := make([]int64, P), make([]uint64, P), make([]uint64, P)
	pending, fromDeleting := make([]uint8, P), make([]uint8, P)
	for i, p := range pods {
		creation[i] = p.CreationTimestamp.Unix()
		uidHi[i], uidLo[i] = uidWords(p.UID)
		pending[i] = lo.Ternary(p.Status.Phase == corev1.PodPending, uint8(1), uint8(0))
		fromDeleting[i] = lo.Ternary(p.Spec.NodeName != "" && s.deletingNodeNames.Has(p.Spec.NodeName), uint8(1), uint8(0))
	}
	desc.n_pods, desc.n_pod_rows = C.uint32_t(P), C.uint32_t(R)
	desc.pod_requests = cI64(a, requests)
	desc.pod_reqs, desc.pod_strict_reqs = podReqs.c(a), strictReqs.c(a)
	desc.pod_tolerates, desc.pod_next_variant = cU64(a, tolerates), cI32(a, next)
	desc.pod_creation, desc.pod_uid_hi, desc.pod_uid_lo = cI64(a, creation), cU64(a, uidHi), cU64(a, uidLo)
	desc.pod_is_pending, desc.pod_from_deleting_node = cU8(a, pending), cU8(a, fromDeleting)

	// ---- topology groups ----
	if err := flattenTopology(ctx, f, s, rows, groups, inverse, taints); err != nil {
		return nil, err
	}

	// ---- options (scheduler.go:103-125) ----
	f.opts.min_values_best_effort = C.uint32_t(lo.Ternary(string(s.minValuesPolicy) == "BestEffort", 1, 0))
	f.opts.max_steps = C.int64_t(maxSteps)
	f.opts.reserved_offering_strict = C.uint32_t(lo.Ternary(s.reservedOfferingMode == ReservedOfferingModeStrict, 1, 0))
	f.opts.reserved_capacity = C.uint32_t(lo.Ternary(karpopts.FromContext(ctx).FeatureGates.ReservedCapacity, 1, 0))
	return f, nil
}

// uidWords: the pod UID as two big-endian words, so that (hi, lo) compares like the UID strings do (queue.go:107).
func uidWords(uid types.UID) (uint64, uint64) {
	s := strings.ReplaceAll(string(uid), "-", "")
	for len(s) < 32 {
		s += "0"
	}
	hi, _ := strconv.ParseUint(s[:16], 16, 64)
	lo_, _ := strconv.ParseUint(s[16:32], 16, 64)
	return hi, lo_
}

// orderedTopologyGroups: Topology.topologyGroups then Topology.inverseTopologyGroups (topology.go:52-56) in a
// deterministic order (Go map iteration is not); the maps are keyed by TopologyGroup.Hash() already, i.e. deduplicated.
func orderedTopologyGroups(t *Topology) (groups, inverse []*TopologyGroup) {
	keys := lo.Keys(t.topologyGroups)
	sort.Slice(keys, func(i, j int) bool { return keys[i] < keys[j] })
	for _, h := range keys {
		groups = append(groups, t.topologyGroups[h])
	}
	keys = lo.Keys(t.inverseTopologyGroups)
	sort.Slice(keys, func(i, j int) bool { return keys[i] < keys[j] })
	for _, h := range keys {
		inverse = append(inverse, t.inverseTopologyGroups[h])
	}
	return groups, inverse
}

// flattenTopology: ksolve_topology. Groups that exist now are `initially_active`; groups that a RELAXED variant of a
// pod would create through Topology.Update (topology.go:162-194) are appended with initially_active = 0 and become real
// on the device when that variant is reached. A pod row owns a group when its UID is in TopologyGroup.owners (for the
// pod as submitted) or when the group was built from that relaxed variant; it is selected by a group when
// TopologyGroup.selects(pod) holds (topologygroup.go:442).
// nolint:gocyclo
func flattenTopology(ctx context.Context, f *flatProblem, s *Scheduler, rows []podRow, groups, inverse []*TopologyGroup, taints []corev1.Taint) error {
	d, a, desc := f.dict, &f.arena, &f.desc
	type entry struct {
		g       *TopologyGroup
		inverse bool
		active  bool
		alias   int32
	}
	var all []entry
	index := map[*TopologyGroup]int{}
	for _, g := range groups {
		index[g] = len(all)
		all = append(all, entry{g: g, active: true, alias: -1})
	}
	for _, g := range inverse {
		index[g] = len(all)
		all = append(all, entry{g: g, inverse: true, active: true, alias: -1})
	}
	// groups a relaxed variant would create: same construction the reference uses (newForTopologies / newForAffinities),
	// looked up by Hash() exactly like Topology.Update does
	P := len(f.pods)
	rowGroups := make([][]int, len(rows))
	byHash := map[uint64]int{}
	for h, g := range s.topology.topologyGroups {
		byHash[h] = index[g]
	}
	aliasOf := map[uint64]int32{}
	for r := P; r < len(rows); r++ {
		affinities, err := s.topology.newForAffinities(ctx, rows[r].pod)
		if err != nil {
			return fmt.Errorf("%w: %v", ErrKSolveUnsupported, err)
		}
		for _, tg := range append(s.topology.newForTopologies(rows[r].pod), affinities...) {
			h := tg.Hash()
			if i, ok := byHash[h]; ok && all[i].active {
				rowGroups[r] = append(rowGroups[r], i)
				continue
			}
			// not there yet: a candidate that comes to exist when this variant is reached. Candidates of one hash built by
			// different pods may differ in what the hash ignores (node-filter values, registered domains): one alias class
			if err := s.topology.countDomains(ctx, tg); err != nil {
				return fmt.Errorf("%w: %v", ErrKSolveUnsupported, err)
			}
			cls, ok := aliasOf[h]
			if !ok {
				cls = int32(len(aliasOf))
				aliasOf[h] = cls
			}
			index[tg] = len(all)
			rowGroups[r] = append(rowGroups[r], len(all))
			all = append(all, entry{g: tg, alias: cls})
		}
	}
	G := len(all)
	if G == 0 {
		return nil
	}
	if G > int(C.KSOLVE_MAX_TOPO_GROUPS) {
		return fmt.Errorf("%w: %d topology groups", ErrKSolveUnsupported, G)
	}
	domainWords := 1
	for _, e := range all {
		if e.g.Key != corev1.LabelHostname {
			k := d.keyIndex[e.g.Key]
			domainWords = max(domainWords, int(d.wordOff[k+1]-d.wordOff[k]))
		}
	}
	E := len(f.nodes)
	typ, inv, act := make([]uint8, G), make([]uint8, G), make([]uint8, G)
	key, skew, minDom, alias := make([]int32, G), make([]int32, G), make([]int32, G), make([]int32, G)
	domains, counts := make([]uint64, G*domainWords), make([]int32, G*domainWords*64)
	nodeCounts := make([]int32, G*max(E, 1))
	affHonor, taintHonor := make([]uint8, G), make([]uint8, G)
	filterFirst := make([]uint32, G+1)
	filterReqs := &reqTable{d: d}
	filterTol := make([]uint64, G)
	nodeIndex := map[string]int{}
	for e, n := range f.nodes {
		nodeIndex[n.HostName()] = e
	}
	for i, e := range all {
		g := e.g
		typ[i] = map[TopologyType]uint8{TopologyTypeSpread: 0, TopologyTypePodAffinity: 1, TopologyTypePodAntiAffinity: 2}[g.Type]
		inv[i], act[i], alias[i] = lo.Ternary(e.inverse, uint8(1), uint8(0)), lo.Ternary(e.active, uint8(1), uint8(0)), e.alias
		skew[i], minDom[i] = g.maxSkew, -1
		if g.minDomains != nil {
			minDom[i] = *g.minDomains
		}
		if g.Key == corev1.LabelHostname {
			key[i] = -1
			for dom, n := range g.domains { // pods pre-counted per existing node (countDomains, topology.go:361-459)
				if e, ok := nodeIndex[dom]; ok {
					nodeCounts[i*max(E, 1)+e] = n
				}
			}
		} else {
			k := d.keyIndex[g.Key]
			key[i] = int32(k)
			for dom, n := range g.domains {
				v := d.valIndex[k][dom]
				domains[i*domainWords+v/64] |= 1 << uint(v%64)
				counts[(i*domainWords)*64+v] = n
			}
		}
		affHonor[i] = lo.Ternary(g.nodeFilter.AffinityPolicy == corev1.NodeInclusionPolicyHonor, uint8(1), uint8(0))
		taintHonor[i] = lo.Ternary(g.nodeFilter.TaintPolicy == corev1.NodeInclusionPolicyHonor, uint8(1), uint8(0))
		filterFirst[i] = uint32(len(filterReqs.defined))
		for _, r := range g.nodeFilter.Requirements {
			filterReqs.add(r)
		}
		probe := &corev1.Pod{Spec: corev1.PodSpec{Tolerations: g.nodeFilter.Tolerations}}
		for ti := range taints {
			if scheduling.Taints([]corev1.Taint{taints[ti]}).ToleratesPod(probe) == nil {
				filterTol[i] |= 1 << uint(ti)
			}
		}
	}
	filterFirst[G] = uint32(len(filterReqs.defined))
	// ---- resident clusters: what NewTopology would have counted had the bound pod rows not been "pods being scheduled" ----
	// s.topology was assembled with every pod of f.pods in excludedPods (topology.go:92-94), so g.domains misses the bound
	// rows. A probe of the sweep takes ITS displaced pods out again; for that the base must count them all (countDomains,
	// topology.go:361-459, restated over the rows), know which domains the NodePools / instance types offer by themselves
	// (domain_universe: ForEachDomain, topologydomaingroup.go:61-72) and how many nodes register each domain
	// (domain_node_regs: the loop over t.stateNodes, topology.go:376-392). The C++ twin: karpenter_amd/host/ksched.cpp.
	var universe []uint64
	var nodeRegs []int32
	if f.boundTo != nil {
		universe, nodeRegs = make([]uint64, G*domainWords), make([]int32, G*domainWords*64)
		nodeByName := map[string]int{}
		for e, n := range f.nodes {
			nodeByName[n.Name()] = e
		}
		nodeReqs := make([]scheduling.Requirements, E)
		for e, n := range f.nodes {
			nodeReqs[e] = scheduling.NewLabelRequirements(n.Labels())
		}
		for i, e := range all {
			g := e.g
			matches := make([]int8, E) // TopologyNodeFilter.Matches per node, evaluated once: 0 unknown, 1 yes, -1 no
			nodeMatches := func(en int) bool {
				if matches[en] == 0 {
					matches[en] = -1
					if g.nodeFilter.Matches(f.nodes[en].Taints(), nodeReqs[en]) {
						matches[en] = 1
					}
				}
				return matches[en] == 1
			}
			if g.Key != corev1.LabelHostname {
				k := d.keyIndex[g.Key]
				probe := &corev1.Pod{Spec: corev1.PodSpec{Tolerations: g.nodeFilter.Tolerations}}
				s.topology.domainGroups[g.Key].ForEachDomain(probe, g.nodeFilter.TaintPolicy, func(dom string) {
					if v, ok := d.valIndex[k][dom]; ok {
						universe[i*domainWords+v/64] |= 1 << uint(v%64)
					}
				})
				for en, n := range f.nodes { // the nodes that register a domain (topology.go:376-392)
					if n.Node == nil || !nodeMatches(en) {
						continue
					}
					if dom, ok := n.Labels()[g.Key]; ok {
						if v, ok := d.valIndex[k][dom]; ok {
							nodeRegs[(i*domainWords)*64+v]++
						}
					}
				}
			}
			if e.inverse {
				continue // inverse groups count their OWNERS' domains: below
			}
			for r, p := range f.pods { // the bound rows this group selects (TopologyListOptions + countDomains' loop)
				if f.boundTo[r] == "" || IgnoredForTopology(p) || !g.selects(p) {
					continue
				}
				en, ok := nodeByName[f.boundTo[r]]
				if !ok {
					continue
				}
				dom, ok := f.nodes[en].Labels()[g.Key]
				if !ok && g.Key == corev1.LabelHostname {
					dom, ok = f.nodes[en].Name(), true
				}
				if !ok || !nodeMatches(en) {
					continue
				}
				if g.Key == corev1.LabelHostname {
					nodeCounts[i*max(E, 1)+en]++
				} else {
					k := d.keyIndex[g.Key]
					v := d.valIndex[k][dom]
					domains[i*domainWords+v/64] |= 1 << uint(v%64)
					counts[(i*domainWords)*64+v]++
				}
			}
		}
		// inverse anti-affinity groups: every bound row that OWNS a required anti-affinity term blocks its node's domain
		// (updateInverseAffinities / updateInverseAntiAffinity with the node's labels, topology.go:310-355)
		for r, p := range f.pods {
			if f.boundTo[r] == "" || IgnoredForTopology(p) || p.Spec.Affinity == nil || p.Spec.Affinity.PodAntiAffinity == nil {
				continue
			}
			en, ok := nodeByName[f.boundTo[r]]
			if !ok {
				continue
			}
			for _, term := range p.Spec.Affinity.PodAntiAffinity.RequiredDuringSchedulingIgnoredDuringExecution {
				namespaces, err := s.topology.buildNamespaceList(ctx, p.Namespace, term.Namespaces, term.NamespaceSelector)
				if err != nil {
					return fmt.Errorf("%w: %v", ErrKSolveUnsupported, err)
				}
				tg := NewTopologyGroup(TopologyTypePodAntiAffinity, term.TopologyKey, p, namespaces, term.LabelSelector, math.MaxInt32, nil, nil, nil, s.topology.domainGroups[term.TopologyKey])
				g, ok := s.topology.inverseTopologyGroups[tg.Hash()]
				if !ok {
					continue
				}
				i := index[g]
				dom, ok := f.nodes[en].Labels()[g.Key]
				if !ok {
					continue
				}
				if g.Key == corev1.LabelHostname {
					nodeCounts[i*max(E, 1)+en]++
				} else {
					k := d.keyIndex[g.Key]
					v := d.valIndex[k][dom]
					domains[i*domainWords+v/64] |= 1 << uint(v%64)
					counts[(i*domainWords)*64+v]++
				}
			}
		}
	}
	// value ranks: ties between domains go to the lexicographically smallest name (the reference's tie is a Go map order)
	rank := make([]uint16, d.reqWords()*64)
	for k := range d.keys {
		order := make([]int, len(d.values[k]))
		for i := range order {
			order[i] = i
		}
		sort.Slice(order, func(x, y int) bool { return d.values[k][order[x]] < d.values[k][order[y]] })
		for r, i := range order {
			rank[int(d.wordOff[k])*64+i] = uint16(r)
		}
	}
	hostValue := make([]int32, max(E, 1))
	for e, n := range f.nodes {
		hostValue[e] = -1
		if k, ok := d.keyIndex[corev1.LabelHostname]; ok {
			if v, ok := d.valIndex[k][n.HostName()]; ok {
				hostValue[e] = int32(v)
			}
		}
	}
	// per pod row: groups owned / selected
	tw := (G + 63) / 64
	owned, selected := make([]uint64, len(rows)*tw), make([]uint64, len(rows)*tw)
	for r, row := range rows {
		for i, e := range all {
			if !e.inverse || r < P { // selection is by namespace + label selector, whatever the variant
				if e.g.selects(row.pod) {
					selected[r*tw+i/64] |= 1 << uint(i%64)
				}
			}
			if r < P {
				if _, ok := e.g.owners[row.pod.UID]; ok {
					owned[r*tw+i/64] |= 1 << uint(i%64)
				}
			}
		}
		for _, i := range rowGroups[r] {
			owned[r*tw+i/64] |= 1 << uint(i%64)
		}
	}
	t := &desc.topo
	t.n = C.uint32_t(G)
	t._type, t.inverse, t.initially_active = cU8(a, typ), cU8(a, inv), cU8(a, act)
	t.key, t.max_skew, t.min_domains = cI32(a, key), cI32(a, skew), cI32(a, minDom)
	t.domain_words = C.uint32_t(domainWords)
	t.domains, t.init_counts = cU64(a, domains), cI32(a, counts)
	if universe != nil {
		t.domain_universe, t.domain_node_regs = cU64(a, universe), cI32(a, nodeRegs)
	}
	if E > 0 {
		t.init_node_counts = cI32(a, nodeCounts)
		t.node_hostname_value = cI32(a, hostValue)
	}
	t.filter_affinity_honor, t.filter_taint_honor = cU8(a, affHonor), cU8(a, taintHonor)
	t.filter_first, t.filter_reqs, t.filter_tolerates = cU32(a, filterFirst), filterReqs.c(a), cU64(a, filterTol)
	t.value_rank = cU16(a, rank)
	if len(aliasOf) > 0 {
		t.alias_class, t.n_alias_classes = cI32(a, alias), C.uint32_t(len(aliasOf))
	}
	desc.pod_topo_owned, desc.pod_topo_selected = cU64(a, owned), cU64(a, selected)
	return nil
}
