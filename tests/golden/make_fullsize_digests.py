#!/usr/bin/env python3
"""Full-size parity pins (TEST INFRASTRUCTURE): runs the ORACLE offline, in the CPU container, on the BASELINE
configurations at the sizes bench.py measures, and commits a digest of the canonical Results under tests/golden/fullsize/.
The `-m gpu` tests and bench.py then solve the same seeded problem on the device and compare digests — the reference's own
in-benchmark gate (scheduling_benchmark_test.go:176-181) made bit-exact.

  python tests/golden/make_fullsize_digests.py config2 1000000 500 42      # ~2 h on one core (O(pods x claims))
  ORACLE_THREADS=128 python tests/golden/make_fullsize_digests.py config3 500000 500 42   # candidate fan-out like parallelizeUntil

The oracle is O(pods x claims); the device is checked against it, never the other way round.
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_problem(name, pods, types, seed, extra):
    from karpenter_amd import fixtures as fx
    if name == "config1":
        return fx.config1(pods=pods, n_types=types, seed=seed)
    if name == "config2":
        return fx.config2(pods=pods, n_types=types, seed=seed)
    if name == "config3":
        return fx.config3(pods=pods, n_types=types, seed=seed, anti_affinity_pods=int(extra) if extra else None)
    if name == "config4":
        return fx.config4(pods=pods, n_types=types, n_pools=int(extra) if extra else 16, seed=seed)
    raise SystemExit("unknown config " + name)


def pin_name(name, pods, types, seed, extra):
    return f"{name}_p{pods}_t{types}_s{seed}" + (f"_x{extra}" if extra else "")


def main():
    name, pods, types, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    extra = sys.argv[5] if len(sys.argv) > 5 else ""
    import oracle
    import parity
    prob = build_problem(name, pods, types, seed, extra)
    t0 = time.time()
    res = oracle.solve(prob)
    dt = time.time() - t0
    digest, fps = parity.results_digest(res)
    out = {"config": name, "pods": pods, "types": types, "seed": seed, "extra": extra,
           "digest": digest, "claims": len(res["newNodeClaims"]), "podErrors": len(res["podErrors"]),
           "binEvaluations": res["counters"]["binEvaluations"], "packingCost": float(res["packingCost"]).hex(),
           "packingCostApprox": res["packingCost"], "oracleSeconds": round(dt, 1), "oracleThreads": int(os.environ.get("ORACLE_THREADS", "1")),
           "claimPods": [len(c["pods"]) for c in res["newNodeClaims"]],
           "claimFingerprints": [f[:12] for f in fps]}
    # PIN_OUT_DIR: where the pin goes when it is made on another machine (the 256-core GPU box: ORACLE_THREADS=128 and the
    # candidate fan-out of the in-flight scan, oracle/scheduler.hpp add_to_inflight_parallel); copied to fullsize/ afterwards
    path = os.path.join(os.environ.get("PIN_OUT_DIR") or os.path.join(HERE, "fullsize"), pin_name(name, pods, types, seed, extra) + ".json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(path, digest, out["claims"], dt)


if __name__ == "__main__":
    main()
