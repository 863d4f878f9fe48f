#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE counter CSVs of tools/pmc_traffic.sh into
profiles/pmc_traffic.json: HBM bytes per launch of every kernel of the bench step.

Units / corrections (MI355X_MICROARCH.md, "HBM"): the counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide coalesced streaming reads (128-B requests tallied at
64 B), so the read side is doubled; WRITE_SIZE is exact for wide stores.  tools/pmc_calibrate.hip
calibrates the patterns these kernels use (dword loads, 32/64-byte row segments): the read side is 1/2
for all of them, a lone dword store is tallied as 32 B — both the raw and the corrected figure are kept.  Only full-batch launches (the most frequent grid size) count."""
import collections, csv, glob, json, sys


def per_kernel(d, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            name = name[5:] if name.startswith("void ") else name          # template instances: "void k_pyr_pad<true>"
            rows[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, v in rows.items():
        if not k.startswith("k_"):
            continue
        big = max(g for g, _ in v)
        vals = [x for g, x in v if g == big]
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
import os
res = {"workload": os.environ.get("PMC_WORKLOAD", "kitti_stereo_1241x376_1000feat"), "batch": int(os.environ.get("PMC_BATCH", "64")),
       "images_per_launch": int(os.environ.get("PMC_IMAGES", "128")),
       "note": "rocprofv3 --pmc, separate passes; FETCH_SIZE doubled per the gfx950 correction", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    res["kernels"][k] = {"fetch_kib_raw": round(f, 1), "write_kib_raw": round(w, 1), "launches_averaged": max(nf, nw),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024), "hbm_bytes_per_launch_uncorrected": int((f + w) * 1024)}
print(json.dumps(res, indent=1))
