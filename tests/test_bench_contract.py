"""bench.py's contract (one JSON line, whole-job value, max-over-ranks timing, summary all-reduce) exercised WITHOUT a
GPU: the launcher command is the driver's (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`), the
collective runs on gloo, and the solver behind the C ABI is the test emulation passed through bench.py's --solver-lib
test hook. Nothing here is a measurement; it pins that the N>1 path runs, that every rank solves its own problem and that
the reduced packing summary is the sum of what the oracle finds for each rank's problem."""
import json
import os
import socket
import subprocess
import sys

import parity
from karpenter_amd import fixtures as fx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _env():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", KSOLVE_BENCH_TEST_HOOK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def test_single_rank_line(oracle):
    emu = parity.build_emu()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--pods", "3000", "--types", "60", "--cpu-sample", "1500", "--cpu-runs", "3",
           "--topology-pods", "400", "--components-pods", "4000", "--components-types", "60", "--components-calibration-pods", "4000", "--whole-batch-pods", "3000", "--whole-batch-exact-pods", "4000", "--beyond-lds-pods", "0", "--batch-problems", "3", "--batch-pods", "400", "--sweep-nodes", "300", "--sweep-candidates", "40", "--sweep-sample", "6", "--sweep-topology-sample", "6", "--sweep-windows", "3", "--sweep-window-size", "12", "--solver-lib", emu]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert REQUIRED <= set(line) and {"cpu_baseline", "batched", "packing", "counters", "parity", "config2_topology", "engine"} <= set(line)
    assert line["engine"] == "cursor" and line["parity"]["oracle_pin"] is None and len(line["parity"]["results_digest"]) == 64
    assert line["config2_topology"]["pods"] == 400 and line["config2_topology"]["value"] > 0 and line["config2_topology"]["engine"] == "spread" and line["config2_topology"]["pack_kernel"]["kernel"] == "ksolve_pack_topo" and "dominant" in line["roofline"]
    assert line["packing"]["per_instance_type"]["launch_types_used"] >= 1
    cc = line["config3_components"]
    assert cc["components"] == 16 and cc["pods"] == 4000 and cc["engines"] == ["cursor"] and abs(cc["calibration"]["cost_rel_delta"]) < 0.05
    assert cc["ranks"] == 1 and cc["per_instance_type"]["claims_from_vector"] == cc["node_claims"]
    assert abs(cc["per_instance_type"]["cost_from_vector"] - cc["packing_cost_per_hour"]) <= 1e-9 * cc["packing_cost_per_hour"]
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["higher_is_better"] is True and line["dtype"] == "int64" and "workload" in line["config"] and "TEST HOOK" in line["data"]
    want = oracle.solve(fx.config2(pods=3000, n_types=60, seed=42))
    assert line["packing"]["pods_scheduled"] == 3000 - len(want["podErrors"]) and line["packing"]["node_claims"] == len(want["newNodeClaims"])
    assert abs(line["packing"]["packing_cost_per_hour"] - want["packingCost"]) <= 1e-9 * want["packingCost"]
    assert line["counters"]["referenceBinEvaluations"] == want["counters"]["binEvaluations"]
    assert abs(line["value"] - line["packing"]["pods_scheduled"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    rf = line["roofline"]            # the streaming kernel of the path (pod classing), not the latency-bound pack kernel
    assert rf["kernel"] == "ksolve_row_hash_coop2" and rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None            # a PMC figure is only quoted for the build and the workload it was measured on
    assert abs(rf["achieved"] - rf["algorithmic_bytes"] / (rf["avg_kernel_ms"] * 1e-3) / 1e9) < 1e-9 * rf["achieved"] and rf["algorithmic_bytes"] == rf["rows"] * rf["bytes_per_row"]
    pk = line["pack_kernel"]
    assert pk["kernel"] == "ksolve_pack_fast" and "latency" in pk["bound"] and pk["traffic"] is None and pk["achieved"] is None and pk["reference_equivalent"]["bytes"] > 0
    assert cc["components_check"]["all_digests_match_oracle"] is True and cc["components_check"]["components"] == 16
    sw = line["config4_sweep"]
    assert sw["nodes"] == 300 and sw["candidates"] == 40 and sum(sw["decisions"].values()) == 40 and sw["unit"] == "probes/s" and sw["value"] > 0
    assert sw["oracle_check"]["all_identical"] is True and sw["oracle_check"]["probes"] >= 3 and sw["cpu_baseline"]["kind"] == "port" and sw["cpu_baseline"]["unit"] == "probes/s"
    assert abs(sw["value"] - sw["candidates"] / sw["seconds"]["library_call"]) <= 1e-9 * sw["value"]
    mn = sw["multi_node"]      # the multi-node half of the replay: every prefix of every window in one sweep, the binary search as a walk
    assert mn["windows"] == 3 and mn["window_candidates"] == 13 and mn["probes"] == 36 == sum(mn["decisions_of_all_prefixes"].values()) and sum(mn["commands"]["decisions"].values()) == 3
    assert mn["oracle_check"]["all_identical"] is True and mn["oracle_check"]["probes"] >= 3 and mn["unit"] == "probes/s" and abs(mn["value"] - mn["probes"] / mn["seconds"]["library_call"]) <= 1e-9 * mn["value"]
    assert line["config2_topology"]["oracle_pin"] is None
    inv = line["config2_topology"]["invariants"]      # no pin at this size: the placements replayed against the reference's topology rules
    assert inv["violations"] == 0 and inv["decisions_replayed"]["pods"] == 400
    assert cc["whole_batch"]["pods"] == 3000 and cc["whole_batch"]["engine"] == "cursor" and cc["whole_batch"]["oracle_pin"] is None
    ex = cc["whole_batch_exact"]
    assert ex["pods"] == 4000 and ex["engine"] == "cursor" and ex["invariants"]["pods_on_claims"] == 4000 and ex["node_claims"] == ex["invariants"]["node_claims"]
    assert ex["packing_cost_per_hour"] > 0 and 0.9 < ex["components_ratio"]["cost"] < 1.1
    assert cc["invariants"]["violations"] == 0 and cc["invariants"]["node_claims_checked"] == cc["node_claims"]
    tw = sw["with_topology_pods"]       # the same sweep over a cluster whose bound pods carry spread constraints, judged by the oracle
    assert tw["nodes"] == 300 and sum(tw["decisions"].values()) == 40 and tw["oracle_check"]["all_identical"] is True and tw["oracle_check"]["probes"] >= 3 and "multi_node" not in tw
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "pods/s" and cb["value"] > 0 and len(cb["runs_seconds"]) == 3
    assert line["batched"]["problems"] == 3 and line["batched"]["value"] > 0


def test_two_ranks_under_the_drivers_launcher(oracle):
    emu = parity.build_emu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--pods", "2500", "--types", "60", "--components-pods", "4000", "--components-types", "60", "--components-calibration-pods", "4000", "--sweep-nodes", "300", "--sweep-candidates", "41", "--sweep-sample", "3", "--sweep-windows", "3", "--sweep-window-size", "12", "--solver-lib", emu]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert REQUIRED <= set(line) and line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert "cpu_baseline" not in line and "batched" not in line           # rank 0 at N=1 only
    want = [oracle.solve(fx.config2(pods=2500, n_types=60, seed=42 + rank)) for rank in range(2)]   # every rank solves its own problem
    assert line["packing"]["pods_scheduled"] == sum(2500 - len(w["podErrors"]) for w in want)
    assert line["packing"]["node_claims"] == sum(len(w["newNodeClaims"]) for w in want)
    cost = sum(w["packingCost"] for w in want)
    assert abs(line["packing"]["packing_cost_per_hour"] - cost) <= 1e-9 * cost
    # whole-job value: pods of ALL ranks over the max-over-ranks time of the timed region
    assert abs(line["value"] - line["packing"]["pods_scheduled"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    # BASELINE configs[3] across the ranks: component c on rank c % 2, per-instance-type (count, $/h) vectors all-reduced — the
    # totals equal the oracle's over the 16 components solved one by one
    from karpenter_amd.components import split_by_nodepool
    cc = line["config3_components"]
    parts = [oracle.solve(sub) for _, sub in split_by_nodepool(fx.config4(pods=4000, n_types=60, n_pools=16, seed=42))]
    assert cc["ranks"] == 2 and cc["components"] == 16 and cc["pods"] == 4000
    assert cc["node_claims"] == sum(len(w["newNodeClaims"]) for w in parts) == cc["per_instance_type"]["claims_from_vector"]
    ccost = sum(w["packingCost"] for w in parts)
    assert abs(cc["packing_cost_per_hour"] - ccost) <= 1e-9 * ccost and abs(cc["per_instance_type"]["cost_from_vector"] - ccost) <= 1e-9 * ccost
    assert abs(cc["calibration"]["cost_rel_delta"]) < 0.05 and abs(cc["calibration"]["claims_delta"]) <= 0.05 * cc["calibration"]["whole_batch"]["node_claims"]
    # BASELINE configs[4] across the ranks: the candidates dealt out round-robin, verdict counts summed with one all-reduce
    sw = line["config4_sweep"]
    assert sw["ranks"] == 2 and sw["candidates_all_ranks"] == 41 and sum(sw["decisions_all_ranks"].values()) == 41 and sw["candidates"] == 21 and sw["value"] > 0
    mn = sw["multi_node"]      # three windows over two ranks: rank 0 holds two of them
    assert mn["ranks"] == 2 and mn["windows"] == 3 and mn["probes"] == 24 and mn["probes_all_ranks"] == 36 and mn["value"] > 0


def test_the_hook_needs_its_environment_switch():
    env = _env()
    env.pop("KSOLVE_BENCH_TEST_HOOK")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pods", "100", "--solver-lib", parity.build_emu()], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "test hook" in r.stderr


def test_no_gpu_and_no_hook_is_loud():
    """Without the test hook a multi-rank launch on a box without GPUs fails instead of computing anything on the CPU."""
    env = dict(_env(), RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pods", "100"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in r.stderr
