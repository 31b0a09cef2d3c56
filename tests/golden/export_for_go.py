"""Writes problems built with karpenter_amd/fixtures.py in the reference's wire shapes, for go/replay_test.go to solve
with the real Go Solve():

    python tests/golden/export_for_go.py /tmp/ksolve-replay            # BASELINE configs, scaled to sizes JSON can carry
    python tests/golden/export_for_go.py /tmp/ksolve-replay 20000      # ... with that many pods for configs[1] / [2]

The Go test writes <name>.result.json files; copy them into tests/golden/go_dump/ and run tests/test_go_dump.py."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import from_go  # noqa: E402
from karpenter_amd import fixtures as fx  # noqa: E402


def problems(pods):
    yield "baseline-config0-5000x50", fx.config1()
    yield f"baseline-config1-{pods}x500", fx.config2(pods=pods, n_types=500, seed=42)
    yield f"baseline-config2-{pods}x500", fx.config3(pods=pods, n_types=500, seed=42, anti_affinity_pods=max(1, pods // 30))


def main():
    out = sys.argv[1]
    pods = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    os.makedirs(out, exist_ok=True)
    for name, prob in problems(pods):
        doc = from_go.to_go(fx.expand_pod_groups(prob), name=name)
        path = os.path.join(out, name + ".json")
        json.dump(doc, open(path, "w"), separators=(",", ":"))
        print(f"{path}: {len(doc['pods'])} pods, {len(doc['instanceTypes'])} instance types, {len(doc['nodePools'])} NodePools")


if __name__ == "__main__":
    main()
