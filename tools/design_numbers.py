"""Markdown summary of a bench.py run for DESIGN.md §7:  python tools/design_numbers.py gpurun_out/<run>/bench_detail.json
(the DETAIL file of the run — round 6 on the stdout line is compact; a round-5 verbose line works too)"""
import json
import sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p, r, rb = j["phases_ms"], j["roofline"], j.get("roofline_build") or {}
print(f"* **configs[1] (headline): {j['value']:.0f} Mrays/s** build + flatten + traverse, {j['ms_per_step']:.4f} ms per step over {j['steps']} steps "
      f"(build {p['build_ms']:.3f}, flatten {p['flatten_ms']:.3f}, traverse {p['traverse_total_ms']:.3f} of which the walk kernel {p['traverse_kernel_ms']:.3f}); "
      f"parity `equal: {str(j.get('parity', {}).get('equal')).lower()}` on all {j.get('parity', {}).get('checked_rays')} rays.")
print(f"  Walk roofline: bound `{r['bound']}` at {r['frac']:.3f} of peak (VALU {r.get('valu_frac')}, LDS {r.get('lds_frac')}, HBM-side {r.get('hbm_frac')}, "
      f"waves waiting {r.get('wait_frac')}); HBM-side traffic {r.get('traffic', 0) / 1e6:.0f} MB per launch against {r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic "
      f"({r['algorithmic_gbs'] / 1e3:.1f} TB/s: above the HBM peak because the working set is LDS- and cache-resident — not a roofline fraction); "
      f"{r['slab_tests_per_s'] / 1e9:.0f} G reference-equivalent slab tests/s.  Builder chain: {rb.get('algorithmic_bytes', 0) / 1e6:.0f} MB algorithmic in {rb.get('ms')} ms = "
      f"{rb.get('frac')} of HBM (a latency chain of {j.get('build_levels')} level launches + 4).")
if j.get("pipelined"):
    print(f"  `pipelined` (one host thread, two streams, never `value`): {j['pipelined']['value']:.0f} Mrays/s.")
x = j.get("step_excludes") or {}
if x and "error" not in x:
    print(f"  Beside the step (each the same step with the thing changed, {x.get('steps')} steps; never `value`; {j.get('settle_steps')} untimed settle steps precede the warmup; "
          f"the step's FlatNode array: {j.get('config', {}).get('flat_array', 'lazy')}): "
          + "; ".join(f"`{k}` {x[k]['value']:.0f} Mrays/s ({x[k]['delta_ms_vs_value'] * 1e3:+.0f} µs)"
                      for k in ("with_ray_gen", "with_flat_array", "lazy_flat_array", "eager_flat_array", "all_arrays_eager", "beside_flat_array", "host_io") if k in x)
          + (f" — host_io moves {sum(x['host_io']['bytes_per_step'].values()) / 1e6:.0f} MB per step at {x['host_io']['pcie_gbs']} GB/s." if "host_io" in x else "."))
    for k, v in ((x.get("host_io") or {}).get("paths") or {}).items():
        print(f"    host_io `{k}`: {v['value']:.0f} Mrays/s, {v['ms_per_step']:.4f} ms per step, {v['pcie_gbs']} GB/s over the link"
              + (f", CSR equal to the pageable path: {str(v.get('csr_equal_to_pageable_path')).lower()}" if "csr_equal_to_pageable_path" in v else ""))
for e in j.get("extra_configs", []):
    if "error" in e:
        print(f"* {e['workload']} {e['dtype']}: ERROR {e['error']}")
        continue
    q, rr = e["phases_ms"], e["roofline"]
    if e.get("harness"):      # the reference's whole bench iteration: ray generation + build + flatten + walk + triangle stage
        base = {"cubes120k": "configs[1]", "standin-primary": "configs[2]"}.get(e["workload"].split("+")[0], e["workload"])
        ch = e.get("cpu_harness") or {}
        print(f"* {base} harness loop `intersect_bh`, {e['harness']}: **{e['value']:.0f} Mrays/s** ({e['ms_per_step']:.3f} ms per step: ray generation {q.get('ray_gen_ms', 0):.3f}, "
              f"build {q['build_ms']:.3f}, flatten {q['flatten_ms']:.3f}, walk + triangle stage {q['traverse_kernel_ms']:.3f}, output {q['traverse_total_ms'] - q['traverse_kernel_ms']:.3f}); "
              f"walk bound `{rr.get('bound')}` {rr.get('frac')}, HBM-side {rr.get('hbm_frac')}; parity equal: {str(e.get('parity', {}).get('equal')).lower()} "
              f"({e.get('parity', {}).get('what')}); the oracle's same loop on {ch.get('cores')} host threads: {ch.get('value')} Mrays/s.")
        continue
    if e.get("harness"):
        continue
    tag = {("standin-primary", "weak"): "configs[2] (10 M primary rays, stand-in scene)", ("standin-incoherent", "weak"): "configs[3], one 12.5 M-ray shard",
           ("standin-incoherent", "strong"): "configs[3] whole (100 M rays on one GPU)", ("cubes120k", "weak"): "configs[4] f64, guide walk",
           ("cubes12m", "weak"): "beyond BASELINE: 12 M triangles, 10 M rays (HBM regime)"}[(e["workload"], e["scaling"])]
    asm = q["traverse_total_ms"] - q["traverse_kernel_ms"] - q.get("ray_convert_ms", 0.0)
    print(f"* {tag}: **{e['value']:.0f} Mrays/s** ({e['ms_per_step']:.3f} ms per step: build {q['build_ms']:.3f}, flatten {q['flatten_ms']:.3f}, walk {q['traverse_kernel_ms']:.3f}, "
          f"CSR assembly {asm:.3f} = {100 * asm / q['traverse_total_ms']:.0f} % of traverse"
          + (f", f32 ray copy {q['ray_convert_ms']:.3f}" if q.get("ray_convert_ms") else "") +
          f"); walk bound `{rr.get('bound')}` {rr.get('frac')}, waiting {rr.get('wait_frac')}, HBM-side {rr.get('hbm_frac')}; parity equal: {str(e.get('parity', {}).get('equal')).lower()}.")
    if "pure_f64_walk" in e:
        f = e["pure_f64_walk"]
        print(f"  Pure f64 walk (every slab test in double precision, `k_traverse_wide<double,…>`): **{f['value']:.0f} Mrays/s** ({f['ms_per_step']:.3f} ms per step, walk "
              f"{f['phases_ms']['traverse_kernel_ms']:.3f}); bound `{f['roofline'].get('bound')}` {f['roofline'].get('frac')}; parity equal: {str(f.get('parity', {}).get('equal')).lower()}.")
c = j.get("cpu_baseline")
if c:
    legs = c.get("legs") or {}
    print(f"* CPU baseline on the same box (oracle, kind `{c['kind']}`, {c['cores']} threads of {c['host_cpus_visible']} visible CPUs): {c['value']:.1f} Mrays/s "
          f"(build {c['build_ms']} ms on {c.get('build_threads')} threads, flatten {c['flatten_ms']} ms, traversal {c['traverse_ms_all_cores']} ms, {c['traverse_ns_per_ray_1thread']} ns/ray single-threaded"
          + (f"; median of the phases {c['value_median']:.1f}; legs " + ", ".join(f"{k} {v.get('value')}" for k, v in legs.items()) + f"; load average {c.get('host_load_1m')}" if legs else "")
          + f") — the GPU step is {j.get('speedup_vs_cpu_baseline')} x; a reported baseline, not a target.")
