#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line: tools/bench_summary.py gpurun_out/x.json [...]"""
import json
import sys
for p in sys.argv[1:]:
    d = json.loads([l for l in open(p) if l.startswith("{")][-1])
    lr = d.get("long_run") or {}
    print("%s: %.1f k frames/s, %.4f ms/step (long run %.1f k), FAST %.4f ms frac %.4f, verified %s" % (
        p, d["value"] / 1e3, d["ms_per_step"], lr.get("value", 0) / 1e3, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["verified"]))
    s = d["stage_ms_per_call"]
    print("   stages alone: pyr %.4f fast %.4f oct %.4f desc %.4f match %.4f" % (s["pyramid"], s["fast"], s["quadtree"], s["describe"], s["stereo_match"]))
    for k, v in (d.get("other_workloads") or {}).items():
        print("   %-42s %8.1f k  %s  verified %s" % (k, v["value"] / 1e3, {a: round(b, 3) for a, b in v["stage_ms_alone"].items()}, v["verified"]))
    for k in ("end_to_end", "tracking_front_end"):
        if k in d:
            print("   %s: %s" % (k, {a: d[k][a] for a in ("value", "ms_per_step", "ms_per_frame", "verified") if a in d[k]}))
    c = d.get("cpu_baseline")
    if c:
        print("   cpu: %.1f frames/s on %d processes; all cores: %s" % (c["value"], c["cores"], c.get("all_cores")))
