"""N>1 path on CPU: world_size-2 gloo. Each rank solves its own independent scheduling problem (one NodePool component
per rank) with the test emulation of the device solver, then the per-instance-type option-count / cost
vector is summed with an all-reduce exactly like bench.py does over RCCL. Rank 0 checks the reduced vector against the
oracle's solution of each shard."""
import json
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, emu, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from karpenter_amd import fixtures as fx
    from karpenter_amd.scheduling import NewScheduler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob = fx.config2(pods=1500, n_types=60, seed=100 + rank)
    res = NewScheduler(prob, solver_lib=emu).Solve()
    names = [t["name"] for t in prob["instanceTypes"]]
    vec = torch.zeros(len(names) + 3, dtype=torch.float64)
    for c in res["newNodeClaims"]:
        for t in c["instanceTypes"]:
            vec[names.index(t)] += 1
    vec[-3], vec[-2], vec[-1] = res["scheduledPods"], res["packingCost"], len(res["newNodeClaims"])
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        q.put({"vec": vec.tolist(), "max": t.item()})
    dist.destroy_process_group()


def test_two_rank_sharded_solve_and_allreduce():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import parity
    from karpenter_amd import fixtures as fx
    emu = parity.build_emu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu, q)) for r in range(2)]
    for p in procs: p.start()
    out = q.get(timeout=240)
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    want_sched = want_cost = want_claims = 0
    hist = None
    for r in range(2):
        prob = fx.config2(pods=1500, n_types=60, seed=100 + r)
        res = oracle.solve(prob)
        names = [t["name"] for t in prob["instanceTypes"]]
        hist = hist or [0] * len(names)
        for c in res["newNodeClaims"]:
            for t in c["instanceTypes"]:
                hist[names.index(t)] += 1
        want_sched += 1500 - len(res["podErrors"]); want_cost += res["packingCost"]; want_claims += len(res["newNodeClaims"])
    assert out["max"] == 2.0
    assert [int(x) for x in out["vec"][:-3]] == hist
    assert int(out["vec"][-3]) == want_sched and int(out["vec"][-1]) == want_claims
    assert abs(out["vec"][-2] - want_cost) < 1e-9 * max(1.0, want_cost)


def _sweep_worker(rank, world, port, emu, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from karpenter_amd import disruption as dz
    from karpenter_amd.scheduling import NewScheduler, SolveBatch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cluster = dz.make_cluster(n_nodes=60, pods_per_node=5, seed=9)          # every rank holds the cluster tables (replicated)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:16]
    mine = list(range(rank, len(cands), world))                              # candidates are sharded, no data-path collective
    cmds = dz.sweep_batched(cluster, [cands[i] for i in mine], lambda ps: SolveBatch([NewScheduler(p, solver_lib=emu) for p in ps]))
    code = {dz.NOOP: 0, dz.DELETE: 1, dz.REPLACE: 2}
    verdicts = torch.full((len(cands),), -1, dtype=torch.int64)
    for i, c in zip(mine, cmds):
        verdicts[i] = code[c["decision"]]
    dist.all_reduce(verdicts, op=dist.ReduceOp.MAX)                          # the (candidate, verdict) all-gather of SURVEY §8e(4)
    if rank == 0:
        q.put(verdicts.tolist())
    dist.destroy_process_group()


def test_two_rank_consolidation_sweep():
    """BASELINE configs[4] shape: the single-node sweep shards its candidates over the ranks (each probe is an independent
    Solve()), the verdicts are gathered, and the first valid candidate in the reference's order is the answer
    (singlenodeconsolidation.go:55-126) — the same one the serial sweep on the oracle finds."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import parity
    from karpenter_amd import disruption as dz
    emu = parity.build_emu()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sweep_worker, args=(r, 2, port, emu, q)) for r in range(2)]
    for p in procs: p.start()
    verdicts = q.get(timeout=240)
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    cluster = dz.make_cluster(n_nodes=60, pods_per_node=5, seed=9)
    cands = dz.sort_candidates(cluster, cluster["nodes"])[:16]
    code = {dz.NOOP: 0, dz.DELETE: 1, dz.REPLACE: 2}
    want = [code[c["decision"]] for c in dz.sweep(cluster, cands, oracle.solve)]
    assert verdicts == want and any(v > 0 for v in verdicts)
    first = next(i for i, v in enumerate(verdicts) if v > 0)
    serial = dz.single_node_consolidation(cluster, cands, oracle.solve)
    assert serial["candidates"] == [cands[first]["name"]]


def test_batch_over_two_devices_through_the_c_abi(oracle):
    """ksolve_solve_batch over handles on different devices (ksolve_options.device): one batch per device, side by side; and the
    per-instance-type (NodeClaim count, $/h) vectors of the solves (ksolve_packing_vector) sum to the whole job's — the
    north_star's reduction as the C ABI offers it to a one-process caller. The emulation has as many "devices" as the options
    name; every problem's Results equal the oracle's, the summed vector equals the one derived from the oracle's claims."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    import parity
    from karpenter_amd import fixtures as fx
    from karpenter_amd.scheduling import NewScheduler, SolveBatch
    emu = parity.build_emu()
    probs = [dict(fx.config2(pods=1500 + 200 * i, n_types=60, seed=50 + i), options={"device": i % 2}) for i in range(5)]
    scheds = [NewScheduler(p, solver_lib=emu) for p in probs]
    got = SolveBatch(scheds)
    n_its = len(probs[0]["instanceTypes"])
    total = np.zeros((n_its, 2))
    want_total = np.zeros((n_its, 2))
    for p, g in zip(probs, got):
        w = oracle.solve(p)
        parity.assert_same_results(g, w)
        for i, c, d in g["packingVector"]:
            total[i] += (c, d)
        want_total += bench.launch_type_vector(p, w)
    assert np.array_equal(total[:, 0], want_total[:, 0]) and np.allclose(total[:, 1], want_total[:, 1], rtol=1e-12, atol=0)
    for s_ in scheds:
        s_.close()


def test_sweep_over_replicas_on_several_devices():
    """ksolve_sweep_replicas (BASELINE configs[4] for a one-process caller that owns several GPUs): three replicas of one resident
    cluster on three "devices" of the emulation, single-node probes and multi-node prefixes dealt out in contiguous shares inside
    the call — decisions, replacements, statuses and reference-equivalent evaluation counts come back in probe order exactly as
    one device produces them, with and without topology constraints on the bound pods."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from karpenter_amd import disruption as dz
    from karpenter_amd.scheduling import NewScheduler
    emu = parity.build_emu()
    for topology in (False, True):
        cc = dz.make_resident_cluster(n_nodes=1500, seed=9, topology=topology)
        scheds = [NewScheduler(dz.compact_problem(dict(cc, options={"device": dev}), pods=[]), solver_lib=emu) for dev in range(3)]
        order = dz.compact_candidates(cc)[:200]
        cands = [[i] for i in order] + [order[:k] for k in range(2, 30)]
        one = scheds[0].Sweep(cands, multi_node=True)
        many = scheds[0].Sweep(cands, multi_node=True, replicas=scheds[1:])
        for k in ("decisions", "allNonPendingPodsScheduled", "claims", "status", "referenceBinEvaluations", "replacements", "reasons"):
            assert one[k] == many[k], (topology, k)
        assert many["timings"]["devices"] == 3 and len(set(one["decisions"])) == 3
        for s in scheds:
            s.close()
