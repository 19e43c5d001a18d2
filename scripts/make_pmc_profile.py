"""profiles/*_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv) and a
kernel-trace stats csv:  python3 scripts/make_pmc_profile.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <kernel_stats.csv> <out.json> "<source note>"
Units per /opt/skills/guides/MI355X_MICROARCH.md: counter values are KiB; on gfx950 FETCH_SIZE reports half of the
coalesced streaming reads (calibrated in r01_v1_pmc_traffic.json on kernels with known bytes), WRITE_SIZE is exact."""
import csv, glob, json, os, sys
from collections import defaultdict
pmc_dir, stats_csv, out, note = sys.argv[1:5]


def per_kernel(counter):
    f = glob.glob(os.path.join(pmc_dir, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    acc, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            acc[k] += float(r["Counter_Value"]); n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


fetch, write = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
avg_ns = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(stats_csv))}
kern = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if k.startswith("__amd") or k not in write:
        continue
    b = int((2.0 * fetch[k] + write[k]) * 1024)
    e = {"FETCH_SIZE_KiB": round(fetch[k], 1), "WRITE_SIZE_KiB": round(write[k], 1), "hbm_bytes_per_launch": b}
    key = k.replace("void ", "")
    if k in avg_ns:
        e["avg_ns"] = avg_ns[k]; e["GBps"] = round(b / avg_ns[k], 1)
    kern[key] = e
json.dump({"source": note, "units": "counter values are KiB; FETCH_SIZE doubled (gfx950 counts 1/2 of coalesced streaming reads; "
           "calibrated in r01_v1_pmc_traffic.json); WRITE_SIZE exact", "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps(kern, indent=1)[:1500])
