"""Prints the numbers DESIGN.md / README.md / profiles/README.md quote from a directory of r03_profile.sh outputs (default: profiles/, prefix r03_)."""
import json, os, re, sys

d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
pre = sys.argv[2] if len(sys.argv) > 2 else "r03_"


def line(name):
    with open(os.path.join(d, pre + name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


for name in ("bench_default", "bench_full_step", "bench_Isaac-Velocity-Flat-Anymal-C-v0", "bench_Isaac-Velocity-Rough-G1-v0", "bench_Isaac-Cartpole-v0",
             "bench_full_step_Isaac-Velocity-Flat-Anymal-C-v0", "bench_full_step_Isaac-Velocity-Rough-G1-v0", "bench_forced_dist_rccl", "bench_rehearsal_2ranks_gloo"):
    try:
        b = line(name + ".json.log")
    except Exception as e:
        print(name, "missing:", e)
        continue
    print(f"{name}: {b['value'] / 1e6:.3f} M env-steps/s, {b['ms_per_step']:.2f} ms, phases {{{', '.join(f'{k}: {v:.2f}' for k, v in b['phase_ms'].items())}}}")
    if "env_step_path" in b:
        print(f"   env_step_path {b['env_step_path']['us_per_env_step_batch']:.1f} us = {b['env_step_path']['value'] / 1e6:.0f} M/s, vs cpu {b['env_step_path'].get('vs_cpu_baseline')}")
    if "env_step_path_full" in b:
        print(f"   env_step_path_full {b['env_step_path_full']['us_per_env_step_batch']:.1f} us = {b['env_step_path_full']['value'] / 1e6:.1f} M/s")
    for key in ("roofline_step", "roofline_producers"):
        for k, v in (b.get(key) or {}).items():
            print(f"   {key}.{k}: {v['avg_launch_us']:.1f} us, {v['achieved']:.0f} GB/s, frac {v['frac']:.3f}")
    if name == "bench_default":
        r = b["roofline"]
        print(f"   roofline: {r['avg_launch_us']:.2f} us, frac {r['frac']:.4f}, traffic {r.get('traffic')}, copy peak {r.get('peak_measured_copy')}")
        print(f"   large_n: {b['roofline_large_n']}")
        m = b["roofline_mfma"]
        print(f"   mfma: frac {m['frac']:.3f} ({m['achieved']:.1f} TF), kernel_only {m.get('kernel_only')}")
        for p in m["per_layer"]:
            print("      ", {k: round(v, 1) if isinstance(v, float) else v for k, v in p.items()})
        print(f"   cpu_baseline {b['cpu_baseline']['value'] / 1e6:.3f} M/s, cores {b['cpu_baseline']['cores']}")
    if "collective" in b and b["collective"]:
        print("   collective", b["collective"].get("backend"), b["collective"].get("allreduce_us"))
for name in ("rocprof_bench_kernel_stats.txt", "rocprof_bench_full_step_kernel_stats.txt", "rocprof_step_kernels_4096.txt", "rocprof_step_kernels_65536.txt"):
    print("==", name)
    for ln in open(os.path.join(d, pre + name)):
        if re.search(r"k_obs|k_term_rew|k_mlp_infer|k_mlp_dw|k_mlp_fwd|k_gae|k_adv|k_action|lstm|orchestrate|k_contact|k_articulation|k_head|reduce_batch", ln):
            f = ln.split()
            print("   ", ln[:60].strip(), f[-6:])
