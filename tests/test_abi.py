"""The C-ABI libraries load and export every symbol include/ksolve.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__
    __graft_entry__.build()
    return True


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ksolve.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ksolve_[a-z_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    fns = declared_functions()
    for f in ("ksolve_create", "ksolve_solve", "ksolve_cancel", "ksolve_results_free", "ksolve_destroy", "ksolve_last_error"):
        assert f in fns


def test_libksolve_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "karpenter_amd", "libksolve.so"))
    for f in declared_functions():
        assert hasattr(lib, f), f
    lib.ksolve_abi_version.restype = ctypes.c_uint32
    assert lib.ksolve_abi_version() == 8
    assert not hasattr(lib, "ksolve_is_emulation")  # the product library is the HIP build, never the test emulation


def test_product_refuses_to_run_without_a_gpu(built):
    """No CPU fallback: on a machine without a gfx950 device the product path raises instead of solving."""
    from karpenter_amd import fixtures as fx
    from karpenter_amd.scheduling import NewScheduler, SolverUnavailable, device_available
    if device_available():
        pytest.skip("a GPU is present")
    with pytest.raises(SolverUnavailable):
        NewScheduler(fx.problem(fx.fake_default_instance_types(), [fx.node_pool()], [fx.pod()])).Solve()


def test_host_library_symbols(built):
    lib = ctypes.CDLL(os.path.join(ROOT, "karpenter_amd", "libksched.so"))
    for f in ("ksched_open", "ksched_solve", "ksched_close", "ksched_error", "ksched_error_kind", "ksched_free", "ksched_solve_json"):
        assert hasattr(lib, f), f


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under karpenter_amd/ or include/ may import, include or link it."""
    bad = []
    for base in ("karpenter_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"^\s*(import|from)\s+oracle\b", text, re.M) or re.search(r"#include\s+\"[^\"]*oracle/", text) or "liboracle" in text:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """cgo (and any other FFI generator) parses include/ksolve.h as C: it must compile as C99 without C++ features, and
    every struct a binding fills must have the same size when compiled as C and as C++ (no hidden padding surprises)."""
    import subprocess
    probe = tmp_path / "probe.c"
    probe.write_text('#include <stdio.h>\n#include "ksolve.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(ksolve_problem_desc), '
                     'sizeof(ksolve_topology), sizeof(ksolve_reqsets), sizeof(ksolve_options), sizeof(ksolve_claims), sizeof(ksolve_results)); return KSOLVE_ABI_VERSION == 8 ? 0 : 1; }\n')
    inc = os.path.join(ROOT, "include")
    sizes = []
    for cc, std, exe in (("gcc", "-std=c99", "probe_c"), ("g++", "-std=c++17", "probe_cpp")):
        src = probe if cc == "gcc" else tmp_path / "probe.cpp"
        if cc == "g++":
            src.write_text(probe.read_text())
        subprocess.check_call([cc, std, "-Wall", "-Werror", "-pedantic", "-I", inc, "-o", str(tmp_path / exe), str(src)])
        sizes.append(subprocess.check_output([str(tmp_path / exe)]).decode().split())
    assert sizes[0] == sizes[1], sizes


def test_go_shim_names_every_entry_point():
    """go/ksolve_shim.go is the binding a Karpenter maintainer adds: it must call the boundary's entry
    points by the names the header declares."""
    shim = open(os.path.join(ROOT, "go", "ksolve_shim.go")).read()
    for f in ("ksolve_create", "ksolve_solve", "ksolve_solve_batch", "ksolve_results_free", "ksolve_destroy", "ksolve_cancel"):
        assert "C." + f in shim, f
    sweep = open(os.path.join(ROOT, "go", "ksolve_sweep.go")).read()        # the consolidation sweep's binding
    for f in ("ksolve_create", "ksolve_sweep", "ksolve_sweep_results_free", "ksolve_last_error"):
        assert "C." + f + "(" in sweep, f
    for t in ("ksolve_sweep_desc", "ksolve_sweep_results"):
        assert "C." + t in sweep, t


EXAMPLE_OUTPUT = "claims=1 [pods=5 its=0x2 cpu=7500 price=0.40] assignment=00000"


def build_example(tmp_path, lib_dir, lib_name, source="ksolve_min.c"):
    """examples/*.c: the C ABI used from plain C (what a cgo shim does), linked against `lib_name` in `lib_dir`."""
    import subprocess
    exe = str(tmp_path / (source[:-2] + "_" + lib_name))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", source), "-o", exe,
                           "-L", lib_dir, "-l" + lib_name, "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_plain_c_example_against_the_emulation(built, tmp_path):
    """The flat problem description filled in by hand in C — no flattener, no Python — gives the expected packing when the
    same ABI is served by the test-only host emulation of the solver."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    emu = parity.build_emu()
    exe = build_example(tmp_path, os.path.dirname(emu), "ksolve_emu")
    assert subprocess.check_output([exe]).decode().strip() == EXAMPLE_OUTPUT


EXAMPLE2_OUTPUT = "claims=1 [pods=2 its=0x1 zone=0x2 cpu=2000 price=0.20] assignment=-2,0,-2,0 errors=0000"


def example2_problem():
    """examples/ksolve_nodes_topology.c built with the Python fixtures instead (for the oracle)."""
    from karpenter_amd import fixtures as fx
    it = fx.fake_instance_type("m", {"cpu": "4", "memory": "8Gi", "pods": "10"},
                               offerings=[fx.offering("on-demand", "zone-a", 0.20), fx.offering("on-demand", "zone-b", 0.20)])
    it["overhead"] = {"cpu": "100m"}
    lab = {"app": "web"}
    pods = [fx.pod(uid=f"00000000-0000-0000-0000-{i + 1:012d}", labels=lab, requests={"cpu": "1", "memory": "512Mi"},
                   topology_spread=[fx.spread(fx.ZONE, lab)]) for i in range(4)]
    node = fx.state_node("node-1", it, "zone-a", used={"cpu": "1900m", "memory": "4Gi", "pods": "5"})
    return fx.problem([it], [fx.node_pool()], pods, state_nodes=[node])


def test_plain_c_example_with_an_existing_node_and_a_topology_group(built, oracle, tmp_path):
    """examples/ksolve_nodes_topology.c: a desc with an existing node and a topology group filled in by hand (what
    go/ksolve_flatten.go emits) against the emulation; the same problem from the fixtures through the oracle gives that packing."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    emu = parity.build_emu()
    exe = build_example(tmp_path, os.path.dirname(emu), "ksolve_emu", "ksolve_nodes_topology.c")
    assert subprocess.check_output([exe]).decode().strip() == EXAMPLE2_OUTPUT
    want = oracle.solve(example2_problem())
    uid = lambda i: f"00000000-0000-0000-0000-{i + 1:012d}"
    assert not want["podErrors"] and len(want["newNodeClaims"]) == 1
    assert want["newNodeClaims"][0]["pods"] == [uid(1), uid(3)]
    assert [e["pods"] for e in want["existingNodes"] if e["pods"]] == [[uid(0), uid(2)]]
    zone = [r for r in want["newNodeClaims"][0]["requirements"] if r["key"] == "topology.kubernetes.io/zone"][0]
    assert zone["operator"] == "In" and zone["values"] == ["zone-b"]


def test_go_binding_uses_only_fields_the_header_declares():
    """go/ksolve_*.go cannot be compiled here (no Go toolchain); what can be checked is that every field of the C structs
    they touch exists in include/ksolve.h under that name (cgo spells the C field `type` as `_type`)."""
    header = open(os.path.join(ROOT, "include", "ksolve.h")).read()
    fields = set(re.findall(r"\b([a-z_0-9]+)\s*(?:\[[^\]]*\])?\s*[;,]", header)) | set(re.findall(r"\*\s*([a-z_0-9]+)\s*[;,]", header))
    used = set()
    for f in ("ksolve_flatten.go", "ksolve_rehydrate.go", "ksolve_shim.go", "ksolve_sweep.go"):
        text = open(os.path.join(ROOT, "go", f)).read()
        assert "//go:build cgo && ksolve" in text and "\npackage scheduling\n" in text, f
        for var in ("desc", "t", "out", "f.opts", "cl", "res", "f.desc", "flat.desc", "rs\\[j\\]"):
            used |= set(re.findall(r"(?<![\w.])" + var + r"\.([a-z_][a-z_0-9]*)\b", text))
    used = {u[1:] if u.startswith("_") else u for u in used}
    go_only = {"add", "c", "free", "flat", "observe", "seal", "key", "value", "close", "handle", "s", "mask", "defined", "complement", "PodErrors", "NewNodeClaims", "ExistingNodes"} - fields
    missing = sorted(u for u in used - fields - go_only if "_" in u or u in ("n", "type", "key", "status", "impl"))
    assert not missing, missing
    consts = set(re.findall(r"C\.(KSOLVE_[A-Z_]+)", "".join(open(os.path.join(ROOT, "go", f)).read() for f in os.listdir(os.path.join(ROOT, "go")))))
    assert consts and all(c in header for c in consts), consts


def test_plain_c_example_refuses_without_a_gpu(built, tmp_path):
    import subprocess
    from karpenter_amd.scheduling import device_available
    if device_available():
        pytest.skip("a GPU is present (tests/test_gpu_parity.py runs the example on it)")
    exe = build_example(tmp_path, os.path.join(ROOT, "karpenter_amd"), "ksolve")
    p = subprocess.run([exe], capture_output=True)
    assert p.returncode == 1 and b"no usable gfx950 device" in p.stderr and not p.stdout


def test_only_tests_smoke_and_bench_touch_the_oracle():
    """oracle/ is the checker: besides tests/, only __graft_entry__ (build + smoke) and bench.py (cpu_baseline) may import it."""
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    bad = []
    for dp, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "__pycache__", "gpurun_out", "oracle", "tests")]
        for f in files:
            path = os.path.join(dp, f)
            if f.endswith((".py", ".sh")) and path not in allowed:
                text = open(path, errors="ignore").read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, re.M) or re.search(r"\bimport\s+[\w, ]*\boracle\b", text) or "liboracle" in text:
                    bad.append(os.path.relpath(path, ROOT))
    assert not bad, bad


def test_go_binding_uses_only_members_the_reference_declares():
    """The Go side of the boundary: go/ksolve_*.go live in the reference's package scheduling and read its unexported state
    (Scheduler.nodeClaimTemplates, Topology.topologyGroups, ExistingNode.remainingResources, ...). They cannot be compiled here
    (no Go toolchain), so — like the header check above — every member they select on those types must exist under that name
    in the reference checkout (struct field, embedded type or method). Skipped where the checkout is absent (the GPU box)."""
    import glob
    ref = "/root/reference/pkg/controllers/provisioning/scheduling"
    if not os.path.isdir(ref):
        pytest.skip("no reference checkout on this machine")
    sources = [open(f).read() for f in glob.glob(ref + "/*.go") if not f.endswith("_test.go")]

    def members(type_name):
        out = set()
        for t in sources:
            m = re.search(r"type %s struct \{(.*?)\n\}" % type_name, t, re.S)
            if m:
                for line in m.group(1).split("\n"):
                    mm = re.match(r"\t(\*?[\w.]+)(\s|$)", line)
                    if mm and not line.strip().startswith("//"):
                        out.add(mm.group(1).lstrip("*").split(".")[-1])      # a field, or the name of an embedded type
            out |= set(re.findall(r"func \(\w+ \*?%s\) (\w+)\(" % type_name, t))
        return out
    text = "".join(open(os.path.join(ROOT, "go", f)).read() for f in ("ksolve_flatten.go", "ksolve_rehydrate.go", "ksolve_shim.go", "ksolve_sweep.go"))
    text = re.sub(r"//[^\n]*", "", text)                                     # comments name things too
    checks = {"Scheduler": r"(?<![\w.])s\.([A-Za-z_]\w*)", "Topology": r"\bs\.topology\.([A-Za-z_]\w*)", "PodData": r"\brow\.data\.([A-Za-z_]\w*)"}
    used_any = 0
    for type_name, pattern in checks.items():
        have = members(type_name)
        assert have, type_name
        used = set(re.findall(pattern, text))
        used_any += len(used)
        missing = sorted(u for u in used if u not in have)
        assert not missing, (type_name, missing)
    # members of ExistingNode / NodeClaim / TopologyGroup the binding reads through loop variables
    for type_name, names in {"ExistingNode": ["remainingResources", "requirements", "cachedTaints", "Pods"], "NodeClaim": ["Pods", "NodeClaimTemplate", "reservedOfferings", "hostname"],
                             "TopologyGroup": ["domains", "emptyDomains", "maxSkew", "minDomains", "nodeFilter", "owners", "Key", "Type"], "Topology": ["topologyGroups", "inverseTopologyGroups"]}.items():
        have = members(type_name)
        for nm in names:
            if re.search(r"\.%s\b" % nm, text):
                assert nm in have, (type_name, nm)
                used_any += 1
    assert used_any > 20
    # accessors the binding adds to pkg/scheduling must not collide with something the reference already has
    sched = "".join(open(f).read() for f in glob.glob("/root/reference/pkg/scheduling/*.go") if not f.endswith("_test.go"))
    assert "func (u *HostPortUsage) Reserved(" not in sched and "func (v *VolumeUsage) Tracked(" not in sched
    assert "reserved " in re.search(r"type HostPortUsage struct \{(.*?)\n\}", sched, re.S).group(1) and "limits " in re.search(r"type VolumeUsage struct \{(.*?)\n\}", sched, re.S).group(1)
