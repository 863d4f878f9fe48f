#!/usr/bin/env python3
"""Markdown kernel tables + traffic lines for profiles/README.md from profiles/<round>_<tag>_kernel_stats.csv / _pmc_traffic.json:
python tools/profiles_readme_tables.py r04 headline kitti2000 fullhd natural"""
import csv, json, os, sys
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rnd = sys.argv[1]
for tag in sys.argv[2:]:
    print("\n%s (`%s_%s_kernel_stats.csv`):\n\n| kernel | calls | avg µs | max µs | %% |\n|---|---|---|---|---|" % (tag, rnd, tag))
    for r in csv.DictReader(open(os.path.join(root, "%s_%s_kernel_stats.csv" % (rnd, tag)))):
        n = r["Name"].split("(")[0].replace("void ", "")
        if n.startswith("k_") and float(r["Percentage"]) > 0.3:
            print("| `%s` | %s | %.1f | %.1f | %.2f |" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
    p = os.path.join(root, "%s_%s_pmc_traffic.json" % (rnd, tag))
    if os.path.exists(p):
        d = json.load(open(p))
        print("\nHBM traffic per launch, MB: " + ", ".join("`%s` %.1f" % (k, v["hbm_bytes_per_launch"] / 1e6) for k, v in sorted(d["kernels"].items())
                                                             if v.get("hbm_bytes_per_launch", 0) > 1e6))
