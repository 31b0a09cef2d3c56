//go:build cgo && ksolve

// ksolve_components.go — the NodePool components of one provisioning batch and their deal over the devices of a node
// (SURVEY.md §8e-1): the Go twin of ksched_split_components (karpenter_amd/host/ksched.cpp), which this repository's tests and
// bench.py drive. The provisioner hands every NodePool and every pending pod to ONE Scheduler
// (pkg/controllers/provisioning/provisioner.go:293-297; the pools in the order of pkg/utils/nodepool/nodepool.go:161-171). When
// the pods pin their NodePool, the batch falls apart into components that never share a NodeClaim:
//
//	two NodePools are in one component when some pod may land on either — its REQUIRED constraints on karpenter.sh/nodepool: a
//	node selector and/or every required node-affinity term carrying In [...] on that key (terms are OR-ed and relaxation only drops
//	terms, preferences.go:38-57, so the union over the terms bounds every relaxed variant) — or when a topology group a pod of one
//	owns (spread constraint, pod affinity / anti-affinity term, topology.go:461-533) selects a pod of the other (its domain counts
//	move with every selected pod that is placed, topology.go:197-224).
//
// Each component is then a Scheduler of its own (NewDeviceScheduler with that component's NodePools and pods, on the device
// SplitComponents assigned it), solved bit-exactly as such; the per-instance-type (NodeClaim count, $/h) vectors are summed with
// ksolve_packing_vector_sum (include/ksolve.h). The union is a packing of equal quality, NOT the reference's pod-for-pod answer for
// the whole batch: the reference re-sorts ALL in-flight NodeClaims with an unstable sort before every scan (scheduler.go:598), so
// which of two equally full NodeClaims of pool A a pod joins depends on where pool B's NodeClaims sit in the array (DESIGN.md §6).
// SplitComponents refuses what it cannot bound: existing nodes, reserved capacity, a pod that is not provably pinned, a selector
// with matchExpressions or a namespaceSelector.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Go toolchain, SURVEY.md §8c).
package scheduling

import (
	"fmt"
	"sort"

	corev1 "k8s.io/api/core/v1"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/controllers/state"
)

// Component is one independent packing problem of the batch.
type Component struct {
	NodePools []*v1.NodePool
	Pods      []*corev1.Pod
	Device    int // index into the devices SplitComponents dealt over
}

// SplitComponents returns the components of (nodePools, pods) in NodePool order, dealt over nDevices by pod count — the
// component with the most pods first, each to the device with the fewest pods so far (LPT: never worse than 4/3 of the best
// possible load) — or an error naming why independence cannot be shown. clusterPods: the pods already bound in the cluster that
// NewTopology counts (topology.go:361-459) — any of them is shared topology state, as in ksched_split_components. A nil LabelSelector
// is treated as selecting every pod of its namespaces here (k8s selects nothing): conservative, it can only merge components.
func SplitComponents(nodePools []*v1.NodePool, pods []*corev1.Pod, stateNodes []*state.StateNode, clusterPods []*corev1.Pod, reservedCapacity bool, nDevices int) ([]Component, error) {
	if len(stateNodes) > 0 {
		return nil, fmt.Errorf("existing nodes are bins every NodePool's pods share")
	}
	if len(clusterPods) > 0 {
		return nil, fmt.Errorf("cluster pods: shared topology counts")
	}
	if reservedCapacity {
		return nil, fmt.Errorf("reservations are shared between NodePools")
	}
	index := map[string]int{}
	for i, np := range nodePools {
		index[np.Name] = i
	}
	parent := make([]int, len(nodePools))
	for i := range parent {
		parent[i] = i
	}
	var find func(int) int
	find = func(x int) int {
		for parent[x] != x {
			parent[x] = parent[parent[x]]
			x = parent[x]
		}
		return x
	}
	first := make([]int, len(pods))
	for i, p := range pods {
		allowed := pinnedPools(p, index, len(nodePools))
		if len(allowed) == 0 {
			return nil, fmt.Errorf("pod %s/%s is not provably pinned to NodePools by its required constraints on %s", p.Namespace, p.Name, v1.NodePoolLabelKey)
		}
		first[i] = allowed[0]
		for _, o := range allowed {
			parent[find(o)] = find(allowed[0])
		}
	}
	// topology groups tie their owner to every pod they select
	for i, p := range pods {
		sels, ok := topologySelectors(p)
		if !ok {
			return nil, fmt.Errorf("pod %s/%s: a topology selector with matchExpressions / a namespaceSelector", p.Namespace, p.Name)
		}
		for _, s := range sels {
			for j, q := range pods {
				if s.namespaces[q.Namespace] && matchesAll(q.Labels, s.match) {
					parent[find(first[j])] = find(first[i])
				}
			}
		}
	}
	compOf := map[int]int{}
	var comps []Component
	for i, np := range nodePools {
		r := find(i)
		if _, ok := compOf[r]; !ok {
			compOf[r] = len(comps)
			comps = append(comps, Component{})
		}
		comps[compOf[r]].NodePools = append(comps[compOf[r]].NodePools, np)
	}
	for i, p := range pods {
		c := compOf[find(first[i])]
		comps[c].Pods = append(comps[c].Pods, p)
	}
	kept := comps[:0]
	for _, c := range comps {
		if len(c.Pods) > 0 {
			kept = append(kept, c)
		}
	}
	if nDevices < 1 {
		nDevices = 1
	}
	order := make([]int, len(kept))
	for i := range order {
		order[i] = i
	}
	sort.SliceStable(order, func(a, b int) bool { return len(kept[order[a]].Pods) > len(kept[order[b]].Pods) })
	load := make([]int, nDevices)
	for _, c := range order {
		best := 0
		for d := 1; d < nDevices; d++ {
			if load[d] < load[best] {
				best = d
			}
		}
		kept[c].Device = best
		load[best] += len(kept[c].Pods)
	}
	return kept, nil
}

// the NodePools a pod can ever land on, as far as its required constraints on karpenter.sh/nodepool say; nil = not provably pinned
func pinnedPools(p *corev1.Pod, index map[string]int, n int) []int {
	have := false
	allowed := make([]bool, n)
	if sel, ok := p.Spec.NodeSelector[v1.NodePoolLabelKey]; ok {
		have = true
		if i, ok := index[sel]; ok {
			allowed[i] = true
		}
	}
	if aff := p.Spec.Affinity; aff != nil && aff.NodeAffinity != nil && aff.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil {
		terms := aff.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution.NodeSelectorTerms
		if len(terms) > 0 {
			union := make([]bool, n)
			pinned := true
			for _, term := range terms {
				inAll := make([]bool, n)
				for i := range inAll {
					inAll[i] = true
				}
				any := false
				for _, q := range term.MatchExpressions {
					if q.Key != v1.NodePoolLabelKey || q.Operator != corev1.NodeSelectorOpIn {
						continue
					}
					any = true
					in := make([]bool, n)
					for _, val := range q.Values {
						if i, ok := index[val]; ok {
							in[i] = true
						}
					}
					for i := range inAll {
						inAll[i] = inAll[i] && in[i]
					}
				}
				if !any { // a term without the pin can reach any pool
					pinned = false
					break
				}
				for i := range union {
					union[i] = union[i] || inAll[i]
				}
			}
			if pinned {
				if !have {
					allowed, have = union, true
				} else {
					for i := range allowed {
						allowed[i] = allowed[i] && union[i]
					}
				}
			}
		}
	}
	if !have {
		return nil
	}
	var out []int
	for i, ok := range allowed {
		if ok {
			out = append(out, i)
		}
	}
	return out
}

type componentSelector struct {
	namespaces map[string]bool
	match      map[string]string
}

// every topology group the pod can own, as (namespaces, matchLabels); false when a term cannot be evaluated here
func topologySelectors(p *corev1.Pod) ([]componentSelector, bool) {
	var out []componentSelector
	// A group on any key but the hostname draws its domain universe from EVERY NodePool (buildDomainGroups, topology.go:104-142) and
	// domainMinCount takes the minimum over all of it (topologygroup.go:300-322): its owner cannot be cut off from the other pools.
	for _, c := range p.Spec.TopologySpreadConstraints {
		if c.TopologyKey != corev1.LabelHostname {
			return nil, false
		}
		if c.LabelSelector == nil {
			out = append(out, componentSelector{map[string]bool{p.Namespace: true}, nil})
			continue
		}
		if len(c.LabelSelector.MatchExpressions) > 0 {
			return nil, false
		}
		out = append(out, componentSelector{map[string]bool{p.Namespace: true}, c.LabelSelector.MatchLabels})
	}
	if aff := p.Spec.Affinity; aff != nil {
		var terms []corev1.PodAffinityTerm
		if aff.PodAffinity != nil {
			terms = append(terms, aff.PodAffinity.RequiredDuringSchedulingIgnoredDuringExecution...)
			for _, w := range aff.PodAffinity.PreferredDuringSchedulingIgnoredDuringExecution {
				terms = append(terms, w.PodAffinityTerm)
			}
		}
		if aff.PodAntiAffinity != nil {
			terms = append(terms, aff.PodAntiAffinity.RequiredDuringSchedulingIgnoredDuringExecution...)
			for _, w := range aff.PodAntiAffinity.PreferredDuringSchedulingIgnoredDuringExecution {
				terms = append(terms, w.PodAffinityTerm)
			}
		}
		for _, t := range terms {
			if t.TopologyKey != corev1.LabelHostname {
				return nil, false
			}
			if t.NamespaceSelector != nil || (t.LabelSelector != nil && len(t.LabelSelector.MatchExpressions) > 0) {
				return nil, false
			}
			ns := map[string]bool{}
			for _, n := range t.Namespaces {
				ns[n] = true
			}
			if len(ns) == 0 {
				ns[p.Namespace] = true
			}
			var match map[string]string
			if t.LabelSelector != nil {
				match = t.LabelSelector.MatchLabels
			}
			out = append(out, componentSelector{ns, match})
		}
	}
	return out, true
}

func matchesAll(labels, match map[string]string) bool {
	for k, v := range match {
		if labels[k] != v {
			return false
		}
	}
	return true
}
