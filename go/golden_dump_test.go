//go:build test_performance

// golden_dump_test.go — records what the REFERENCE Solve() does, for pinning the MI355X solver's CPU oracle to the actual Go
// implementation (SURVEY.md §8(f)-1). It is not compiled in the solver's repository (no Go toolchain there); a maintainer
// copies it next to pkg/controllers/provisioning/scheduling/scheduling_benchmark_test.go (same package, same build tag:
// it reuses that file's pod generators) and runs
//
//	KSOLVE_DUMP_DIR=/tmp/ksolve-dump go test -tags=test_performance -run TestDumpGolden ./pkg/controllers/provisioning/scheduling/
//
// Each dump is one JSON document with the inputs in their Kubernetes / Karpenter wire shapes (corev1.Pod, v1.NodePool,
// the instance types flattened to plain structs) and the outputs of Solve(): NodeClaims in the order of
// Results.NewNodeClaims with their pods in commit order, instance type options in order, requirements and requests, and
// the pod errors. In the solver's repository `tests/golden/from_go.py` converts a dump into the problem format of
// `karpenter_amd/fixtures.py`, and `tests/test_go_dump.py` checks the oracle against every dump found under
// tests/golden/go_dump/ (claim by claim, pod by pod — the same L1-strict comparison the device is held to).
package scheduling_test

import (
	"context"
	"encoding/json"
	"fmt"
	"os"
	"path/filepath"
	"testing"

	corev1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/api/resource"
	"k8s.io/apimachinery/pkg/util/sets"
	"k8s.io/client-go/tools/record"
	"k8s.io/utils/clock"
	fakecr "sigs.k8s.io/controller-runtime/pkg/client/fake"

	v1 "sigs.k8s.io/karpenter/pkg/apis/v1"
	"sigs.k8s.io/karpenter/pkg/cloudprovider"
	"sigs.k8s.io/karpenter/pkg/cloudprovider/fake"
	"sigs.k8s.io/karpenter/pkg/controllers/provisioning/scheduling"
	"sigs.k8s.io/karpenter/pkg/controllers/state"
	"sigs.k8s.io/karpenter/pkg/events"
	"sigs.k8s.io/karpenter/pkg/operator/injection"
	"sigs.k8s.io/karpenter/pkg/operator/options"
	"sigs.k8s.io/karpenter/pkg/test"
)

type dumpOffering struct {
	Requirements        []v1.NodeSelectorRequirementWithMinValues `json:"requirements"`
	Price               float64                                   `json:"price"`
	Available           bool                                      `json:"available"`
	ReservationCapacity int                                       `json:"reservationCapacity"`
}

type dumpInstanceType struct {
	Name         string                                    `json:"name"`
	Requirements []v1.NodeSelectorRequirementWithMinValues `json:"requirements"`
	Offerings    []dumpOffering                            `json:"offerings"`
	Capacity     corev1.ResourceList                       `json:"capacity"`
	Overhead     corev1.ResourceList                       `json:"overhead"`
}

type dumpClaim struct {
	NodePool      string                                    `json:"nodePool"`
	Pods          []string                                  `json:"pods"`
	InstanceTypes []string                                  `json:"instanceTypes"`
	Requirements  []v1.NodeSelectorRequirementWithMinValues `json:"requirements"`
	Requests      corev1.ResourceList                       `json:"requests"`
}

type dumpResults struct {
	NewNodeClaims []dumpClaim       `json:"newNodeClaims"`
	PodErrors     map[string]string `json:"podErrors"`
}

type dumpDocument struct {
	Name             string             `json:"name"`
	PreferencePolicy string             `json:"preferencePolicy"`
	WellKnownLabels  []string           `json:"wellKnownLabels"`
	NodePools        []*v1.NodePool     `json:"nodePools"`
	InstanceTypes    []dumpInstanceType `json:"instanceTypes"`
	Pods             []*corev1.Pod      `json:"pods"`
	Results          dumpResults        `json:"results"`
}

func flattenInstanceType(it *cloudprovider.InstanceType) dumpInstanceType {
	out := dumpInstanceType{
		Name:         it.Name,
		Requirements: it.Requirements.NodeSelectorRequirements(),
		Capacity:     it.Capacity,
		Overhead:     it.Overhead.Total(),
	}
	for _, of := range it.Offerings {
		out.Offerings = append(out.Offerings, dumpOffering{
			Requirements:        of.Requirements.NodeSelectorRequirements(),
			Price:               of.Price,
			Available:           of.Available,
			ReservationCapacity: of.ReservationCapacity,
		})
	}
	return out
}

func dumpOne(t *testing.T, dir, name string, instanceTypeCount int, pods []*corev1.Pod, ignorePreferences bool) {
	dumpCtx := options.ToContext(injection.WithControllerName(context.Background(), "provisioner"), test.Options())
	nodePool := test.NodePool(v1.NodePool{
		Spec: v1.NodePoolSpec{
			Limits: v1.Limits{
				corev1.ResourceCPU:    resource.MustParse("10000000"),
				corev1.ResourceMemory: resource.MustParse("10000000Gi"),
			},
		},
	})
	provider := fake.NewCloudProvider()
	instanceTypes := fake.InstanceTypes(instanceTypeCount)
	provider.InstanceTypes = instanceTypes
	kube := fakecr.NewFakeClient()
	clk := &clock.RealClock{}
	clusterState := state.NewCluster(clk, kube, provider)
	var opts []scheduling.Options
	if ignorePreferences {
		opts = append(opts, scheduling.IgnorePreferences)
	}
	byPool := map[string][]*cloudprovider.InstanceType{nodePool.Name: instanceTypes}
	topology, err := scheduling.NewTopology(dumpCtx, kube, clusterState, nil, []*v1.NodePool{nodePool}, byPool, pods, opts...)
	if err != nil {
		t.Fatalf("creating topology, %s", err)
	}
	scheduler := scheduling.NewScheduler(dumpCtx, kube, []*v1.NodePool{nodePool}, clusterState, nil, topology, byPool, nil,
		events.NewRecorder(&record.FakeRecorder{}), clk, nil, nil, opts...)
	results, err := scheduler.Solve(dumpCtx, pods)
	if err != nil {
		t.Fatalf("solving %s, %s", name, err)
	}

	doc := dumpDocument{
		Name:             name,
		PreferencePolicy: map[bool]string{false: "Respect", true: "Ignore"}[ignorePreferences],
		WellKnownLabels:  sets.List(v1.WellKnownLabels),
		NodePools:        []*v1.NodePool{nodePool},
		Pods:             pods,
		Results:          dumpResults{PodErrors: map[string]string{}},
	}
	for _, it := range instanceTypes {
		doc.InstanceTypes = append(doc.InstanceTypes, flattenInstanceType(it))
	}
	for _, nc := range results.NewNodeClaims {
		claim := dumpClaim{
			NodePool:     nc.NodePoolName,
			Requirements: nc.Requirements.NodeSelectorRequirements(),
			Requests:     nc.Spec.Resources.Requests,
		}
		for _, p := range nc.Pods {
			claim.Pods = append(claim.Pods, string(p.UID))
		}
		for _, it := range nc.InstanceTypeOptions {
			claim.InstanceTypes = append(claim.InstanceTypes, it.Name)
		}
		doc.Results.NewNodeClaims = append(doc.Results.NewNodeClaims, claim)
	}
	for p, podErr := range results.PodErrors {
		doc.Results.PodErrors[string(p.UID)] = podErr.Error()
	}
	raw, err := json.Marshal(doc)
	if err != nil {
		t.Fatalf("encoding %s, %s", name, err)
	}
	path := filepath.Join(dir, name+".json")
	if err := os.WriteFile(path, raw, 0o644); err != nil {
		t.Fatalf("writing %s, %s", path, err)
	}
	fmt.Printf("%s: %d pods -> %d NodeClaims, %d pod errors -> %s\n", name, len(pods), len(results.NewNodeClaims), len(results.PodErrors), path)
}

func TestDumpGolden(t *testing.T) {
	dir := os.Getenv("KSOLVE_DUMP_DIR")
	if dir == "" {
		t.Skip("KSOLVE_DUMP_DIR is not set")
	}
	if err := os.MkdirAll(dir, 0o755); err != nil {
		t.Fatalf("creating %s, %s", dir, err)
	}
	// the reference's own benchmark shapes (scheduling_benchmark_test.go): generic pods, the diverse mix with topology
	// spread / affinity / anti-affinity, and pods with preferences under both preference policies
	dumpOne(t, dir, "generic-2000x400", 400, makeGenericPods(2000), false)
	dumpOne(t, dir, "diverse-500x400", 400, makeDiversePods(500), false)
	dumpOne(t, dir, "diverse-5000x400", 400, makeDiversePods(5000), false)
	dumpOne(t, dir, "preference-1000x400-respect", 400, makePreferencePods(1000), false)
	dumpOne(t, dir, "preference-1000x400-ignore", 400, makePreferencePods(1000), true)
}
