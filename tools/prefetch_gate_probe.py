"""Is the pyramid built ahead really ordered behind the FAST stage?  From a rocprofv3 --kernel-trace --hip-runtime-trace run of
bench.py: per step, the host time of the hipStreamWaitEvent / first pad launch against the GPU start of that pad and the GPU end
of the FAST kernel in front of it.   python tools/prefetch_gate_probe.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
kt = sorted(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
api = sorted(csv.DictReader(open(glob.glob(d + "/*/*hip_api_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
print(api[0].keys())
pads = [r for r in kt if "k_pyr_pad" in r["Kernel_Name"]]
fasts = [r for r in kt if "k_fast_strips" in r["Kernel_Name"]]
corr = {r["Correlation_Id"]: r for r in api}
t0 = int(pads[-6]["Start_Timestamp"])
for p in pads[-6:-1]:
    ps = int(p["Start_Timestamp"])
    host = corr.get(p["Correlation_Id"])
    f_before = [f for f in fasts if int(f["Start_Timestamp"]) < ps][-1]
    print("pad gpu start %9.1f | host launch %9.1f (%s) | FAST before it: start %9.1f end %9.1f" % (
        (ps - t0) / 1e3, (int(host["Start_Timestamp"]) - t0) / 1e3 if host else -1, host["Function"] if host else "?",
        (int(f_before["Start_Timestamp"]) - t0) / 1e3, (int(f_before["End_Timestamp"]) - t0) / 1e3))
# host calls around the last-but-two pad
p = pads[-3]
h = corr.get(p["Correlation_Id"])
if h:
    hs = int(h["Start_Timestamp"])
    for r in api:
        s = int(r["Start_Timestamp"])
        if hs - 150000 < s < hs + 100000:
            print("   host %9.1f %s" % ((s - t0) / 1e3, r["Function"]))
