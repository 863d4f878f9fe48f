#!/usr/bin/env python3
"""Counter CSVs of tools/pmc_calibrate.sh -> counted bytes / touched bytes per access pattern."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in rows.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
exp = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
res = {"note": "counter KiB x 1024 / bytes touched once (1 GiB buffer, past the 256 MiB Infinity Cache)", "patterns": {}}
for k in ("cal_read16", "cal_read4", "cal_tile<64>", "cal_tile<32>", "cal_window"):
    res["patterns"][k] = {"touched_bytes": exp[k], "FETCH_SIZE_bytes": int(fetch.get(k, 0) * 1024),
                          "ratio": round(fetch.get(k, 0) * 1024 / exp[k], 4)}
res["patterns"]["cal_write16"] = {"touched_bytes": exp["cal_write16"], "WRITE_SIZE_bytes": int(write.get("cal_write16", 0) * 1024),
                                  "ratio": round(write.get("cal_write16", 0) * 1024 / exp["cal_write16"], 4)}
w = write.get("cal_write4s", 0) * 1024
res["patterns"]["cal_write4s"] = {"dwords_written_bytes": exp["cal_write4s_dwords"], "lines64_touched_bytes": exp["cal_write4s_lines64"],
                                  "WRITE_SIZE_bytes": int(w), "ratio_to_dwords": round(w / exp["cal_write4s_dwords"], 4),
                                  "ratio_to_64B_lines": round(w / exp["cal_write4s_lines64"], 4)}
print(json.dumps(res, indent=1))
