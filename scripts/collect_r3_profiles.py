#!/usr/bin/env python3
"""Copy the round-3 measurement set from gpurun_out/r3f_* (scripts/r3_measure.sh) into profiles/ and build
profiles/r03_pmc.json (HBM traffic per launch of every configuration's kernels: FETCH_SIZE doubled as the microarch
guide prescribes for gfx950, WRITE_SIZE as counted)."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def copy(src, dst):
    if os.path.exists(os.path.join(OUT, src)):
        shutil.copyfile(os.path.join(OUT, src), os.path.join(PROF, dst))


copy("r3f_summary.txt", "r03_cfgA_rocprofv3_summary.txt")
copy("r3f_bench_cfgA.json", "r03_bench_cfgA.json")
copy("r3f_bench_cfgA_driver.json", "r03_bench_cfgA_driver_steps20.json")
copy("r3f_next_rows.jsonl", "r03_next_rows.jsonl")
copy("r3f_readme_shapes.jsonl", "r03_readme_shapes.jsonl")
copy("r3f_sweep.jsonl", "r03_sweep_vs_rocfft.jsonl")
copy("r3f_f64.txt", "r03_float64.txt")


def parse_summary(path):
    """-> {kernel name: {"avg_ns": .., "FETCH_SIZE": KiB, "WRITE_SIZE": KiB}}"""
    res = {}
    if not os.path.exists(path):
        return res
    for line in open(path):
        if line.startswith("void fc::") and line.count(", ") >= 7 and "mean=" not in line:
            parts = line.rstrip().rsplit(", ", 7)          # name (may hold commas, may be cut), calls, total, avg, %, min, max, std
            try:
                if int(parts[1]) > 5:
                    res.setdefault(parts[0][:58].strip(), {})["avg_ns"] = float(parts[3])
            except ValueError:
                pass
            continue
        m = re.match(r"^(void fc::.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([\d.e+]+)", line)
        if m and int(m.group(3)) > 5:
            res.setdefault(m.group(1)[:58].strip(), {})[m.group(2)] = float(m.group(4))
    return res


entries = []
others = []
for cfg, summ in (("cfgA", "r3f_summary.txt"), ("cfg0", "r3f_cfg0_summary.txt"), ("cfgB", "r3f_cfgB_summary.txt"),
                  ("cfgC", "r3f_cfgC_summary.txt"), ("cfgD", "r3f_cfgD_summary.txt"), ("cfgA_shard", "r3f_cfgA_shard_summary.txt"),
                  ("cfgB_shard", "r3f_cfgB_shard_summary.txt"), ("cfgC_shard", "r3f_cfgC_shard_summary.txt")):
    ks = parse_summary(os.path.join(OUT, summ))
    total, per = 0.0, {}
    for name, v in ks.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            per[name] = {"FETCH_SIZE_KiB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"], "traffic_bytes": int(b),
                         "rocprofv3_avg_kernel_ns": v.get("avg_ns")}
            total += b
    if not per:
        continue
    dom = max(per, key=lambda k: per[k]["rocprofv3_avg_kernel_ns"] or 0)
    entries.append({"workload": cfg, "kernel": dom, "traffic_bytes_per_launch": int(per[dom]["traffic_bytes"]),
                    "whole_forward_traffic_bytes": int(total), "kernels": per,
                    "source": f"gpurun_out/{summ} -> profiles/r03_other_configs_traffic.txt" if cfg != "cfgA" else "profiles/r03_cfgA_rocprofv3_summary.txt",
                    "correction": "read bytes = 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request on wide coalesced reads, "
                                  "MI355X_MICROARCH.md); WRITE_SIZE as counted"})
    if cfg != "cfgA":
        bench = os.path.join(OUT, f"r3f_bench_{cfg}.json")
        others.append(f"######## {cfg}: scripts/gpu_prof_traffic.sh r3f_{cfg} --config {cfg}  (kernel trace + FETCH_SIZE + WRITE_SIZE passes, "
                      f"KiB per launch; the JSON line is bench.py --config {cfg} --steps 100 --warmup 20 on the same box without the profiler)")
        if os.path.exists(bench):
            others.append(open(bench).read().strip())
        others.append(open(os.path.join(OUT, summ)).read().strip())
        others.append("")
with open(os.path.join(PROF, "r03_pmc.json"), "w") as fh:
    json.dump({"round": 3, "entries": entries}, fh, indent=1)
with open(os.path.join(PROF, "r03_other_configs_traffic.txt"), "w") as fh:
    fh.write("\n".join(others) + "\n")
for e in entries:
    print(e["workload"], e["kernel"][:50], e["traffic_bytes_per_launch"], e["whole_forward_traffic_bytes"])
