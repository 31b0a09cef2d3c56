"""Shared helpers for parity tests: canonical comparison of a product Results document with the oracle's."""
import hashlib
import json
import os

EMU_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libksolve_emu.so")


def build_emu(reverse_lanes=False):
    """Builds the TEST-ONLY host emulation of the device solver (tests/emu/ksolve_emu.cpp). reverse_lanes: the variant
    whose wave-wide calls run their lanes in the opposite order (csrc/wave.h, KS_EMU_REVERSE_LANES) — same answers expected."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "emu", "ksolve_emu.cpp")
    deps = [src] + [os.path.join(root, "karpenter_amd", "csrc", f) for f in os.listdir(os.path.join(root, "karpenter_amd", "csrc")) if f.endswith(".h")]
    deps.append(os.path.join(root, "include", "ksolve.h"))
    lib = EMU_LIB.replace(".so", "_reversed.so") if reverse_lanes else EMU_LIB
    stale = lambda: not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps)
    if stale():
        import fcntl
        with open(lib + ".lock", "w") as lock:      # pytest-xdist workers: one of them builds, the others wait and find it fresh
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                tmp = lib + f".{os.getpid()}.tmp"
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread"] + (["-DKS_EMU_REVERSE_LANES"] if reverse_lanes else []) + ["-o", tmp, src])
                os.replace(tmp, lib)
    return lib


HOOKS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libksolve_hooks.so")


def build_hooks():
    """The DEVICE library compiled once more with -DKSOLVE_TEST_HOOKS (tests/emu/libksolve_hooks.so): the only gfx950 build that
    reads the KSOLVE_TEST_* / KSOLVE_ROWHASH_KERNEL switches (narrowed row hash, deterministic cancellation, the previous classing
    kernels). karpenter_amd/libksolve.so — the product — is compiled without them. Built here when missing or stale (several
    minutes of hipcc), so keep it built in-tree before a gpurun call: it travels with the snapshot."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import __graft_entry__ as ge
    ge.build_ksolve(HOOKS_LIB, defines=("-DKSOLVE_TEST_HOOKS",))
    return HOOKS_LIB


def canon_req(r):
    return (r["key"], r["complement"], tuple(sorted(r["values"])), r["gte"], r["lte"], r["minValues"])


def canon_claim(c):
    reqs = tuple(sorted(canon_req(r) for r in c["requirements"]))
    requests = tuple(sorted((k, int(v)) for k, v in c["requests"].items() if int(v) != 0))
    return {"nodePool": c["nodePool"], "pods": list(c["pods"]), "instanceTypes": list(c["instanceTypes"]), "requirements": reqs,
            "requests": requests, "hostname": c["hostname"]}


def assert_same_results(got, want, check_price=True):
    """L1-strict parity: same claims in the same order, same pod identities in the same slot order, same instance
    type options, requirements and requests; same existing-node assignments; same pod errors."""
    assert len(got["newNodeClaims"]) == len(want["newNodeClaims"]), (len(got["newNodeClaims"]), len(want["newNodeClaims"]))
    for i, (g, w) in enumerate(zip(got["newNodeClaims"], want["newNodeClaims"])):
        cg, cw = canon_claim(g), canon_claim(w)
        for field in cw:
            assert cg[field] == cw[field], (i, field, cg[field], cw[field])
        if check_price:
            assert g["cheapestPrice"] == w["cheapestPrice"], (i, g["cheapestPrice"], w["cheapestPrice"])
    ge = {e["name"]: e["pods"] for e in got.get("existingNodes", []) if e["pods"]}
    we = {e["name"]: e["pods"] for e in want.get("existingNodes", []) if e["pods"]}
    assert ge == we
    gerr = {u: (e["code"], e["diag"]) for u, e in got["podErrors"].items()}
    werr = {u: (e["code"], e["diag"]) for u, e in want["podErrors"].items()}
    assert gerr == werr, (sorted(gerr.items())[:5], sorted(werr.items())[:5])


def claim_fingerprint(c):
    """sha256 of one NodeClaim in canonical form (pod identities in slot order, instance-type options, requirements,
    requests, hostname, NodePool, the cheapest launch price bit-for-bit)."""
    cc = canon_claim(c)
    doc = [cc["nodePool"], cc["hostname"], cc["pods"], cc["instanceTypes"], cc["requirements"], cc["requests"], float(c["cheapestPrice"]).hex()]
    return hashlib.sha256(json.dumps(doc, separators=(",", ":")).encode()).hexdigest()


def results_digest(res):
    """Digest of a whole Results document at L1-strict level: the claim stream in order, existing-node assignments and
    pod errors. Two documents with the same digest pass assert_same_results. Used to pin the device at sizes where the
    oracle takes hours: the oracle runs offline (tests/golden/make_fullsize_digests.py), the digest is committed."""
    h = hashlib.sha256()
    fps = [claim_fingerprint(c) for c in res["newNodeClaims"]]
    for f in fps:
        h.update(f.encode())
    for e in sorted(res.get("existingNodes", []), key=lambda e: e["name"]):
        if e["pods"]:
            h.update(json.dumps([e["name"], e["pods"]], separators=(",", ":")).encode())
    for u, e in sorted(res["podErrors"].items()):
        h.update(json.dumps([u, e["code"], e["diag"]], separators=(",", ":")).encode())
    return h.hexdigest(), fps
