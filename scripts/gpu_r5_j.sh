# round 5, GPU pass J: the build with the one-wavefront kernel as the default again (loop with one exit), topology group state in LDS
# in the general engine, class slots clustered by template (four-row problems skip the rows a claim's template cannot meet), the
# sweep through several handles of one device — pins first, then the whole default bench line
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
for eng in auto cursor-pair cursor-wide cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config2_p1000000_t500_s42.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
for eng in auto cursor-hbm; do timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config4_p1000000_t1000_s42_x16.json $eng 2>&1 | tail -1 | tee -a $O/pins.log; done
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config3_p200000_t500_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 300 python tests/tools/gpu_check_pin.py tests/golden/fullsize/config1_p5000_t50_s42.json auto 2>&1 | tail -1 | tee -a $O/pins.log
timeout 1500 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5j/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
print("pack", {k: d["pack_kernel"].get(k) for k in ("avg_kernel_ms", "us_per_pod")})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_full_size", {}).get("value"), d.get("cpu_baseline_engine_host"))
print("e2e", d["end_to_end"]["pods_per_s_through_the_boundary"])
t = d.get("config2_topology", {}); print("topology", t.get("seconds"), t.get("pack_kernel_ms"), t.get("oracle_pin"))
b = d.get("config1_beyond_lds", {}); print("beyond", b.get("seconds"), b.get("value"), b.get("oracle_pin"))
c = d.get("config3_components", {}); print("components", {k: c.get(k) for k in ("seconds", "value")}); print("exact", c.get("whole_batch_exact")); print("whole 1M", c.get("whole_batch"))
s = d.get("config4_sweep", {}); print("sweep", s.get("seconds"), s.get("value"), s.get("oracle_pin")); print("contexts", s.get("contexts_on_one_device")); print("multi", (s.get("multi_node") or {}).get("seconds"), (s.get("multi_node") or {}).get("oracle_pin")); print("topo sweep", (s.get("with_topology_pods") or {}).get("value"), (s.get("with_topology_pods") or {}).get("oracle_pin"))
print("batched", d.get("batched"))
PY
