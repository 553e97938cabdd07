#!/usr/bin/env python3
"""Summarises the bench lines under gpurun_out/ (written by tools/gpu_bench_profiles.sh) into profiles/archive/round2_bench_configs.txt."""
import glob, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = ["bench.log", "bench_k20.log", "bench_c2.log", "bench_c4.log", "bench_c5.log", "bench_n32768.log", "bench_n131072.log",
         "bench_gpus2_shared.log", "bench_c4_gpus2_shared.log"]
out = ["Round-2 bench legs on one MI355X (gpurun box), `python bench.py --config ...` (hipGraph-replayed regions clocked by stream events",
       "unless noted); one line per run: config, n_gpus, value, ms/step, kernel us per step (HIP events), algorithmic GB/s, extras", ""]
for n in names:
    p = os.path.join(R, "gpurun_out", n)
    if not os.path.exists(p):
        continue
    lines = [l for l in open(p) if l.startswith("{")]
    if not lines:
        out.append(f"{n:28s} (no JSON line)")
        continue
    d = json.loads(lines[-1]); r = d["roofline"]; c = d["config"]; t = d["timing"]
    out.append(f"{n:28s} {c['id']} envs/GPU {c['envs_per_gpu']:6d} n_gpus {d['n_gpus']} steps {d['steps']:3d} value {d['value']/1e6:8.1f} M env-steps/s  "
               f"ms/step {d['ms_per_step']:.5f}  kernel {r['avg_launch_us']:.2f} us  {r['achieved']:.0f} GB/s ({r['frac']:.3f} of 8 TB/s)  "
               f"[{'hipGraph' if t['launch'] != 'eager' else 'eager'}; {c['parallelism']}]")
    if "host_region_ms" in t:
        out.append(f"    host-clock regions (ms): {t['host_region_ms'][:5]}  device-clock regions (ms): {t['region_ms'][:5]}")
    if d.get("extras"):
        out.append("    extras: " + json.dumps(d["extras"]))
    if d.get("floodfill"):
        out.append("    floodfill: " + json.dumps(d["floodfill"]))
    cb = d.get("cpu_baseline")
    if cb:
        ns = cb.get("numpy_step", {})
        out.append(f"    cpu_baseline (host: {cb.get('host_cores_available')} cores): oracle C 1 thread {cb['value']/1e6:.2f} M/s, "
                   f"{cb.get('all_cores', {}).get('cores')} threads {cb.get('all_cores', {}).get('value', 0)/1e6:.1f} M/s; NumPy per-env step() 1 process "
                   f"{ns.get('value', 0)/1e3:.1f} k/s, {ns.get('all_cores', {}).get('cores')} processes {ns.get('all_cores', {}).get('value', 0)/1e3:.0f} k/s")
out += ["", "(bench_gpus2_shared / bench_c4_gpus2_shared: `--gpus 2` on a ONE-GPU box — the two self-spawned ranks share the GPU and use a gloo",
        "control plane; a functional check of the N>1 code path, not a scaling number.  c4 with one rank: step + packing launch per step.)"]
open(os.path.join(R, "profiles", "round2_bench_configs.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
