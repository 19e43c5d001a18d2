"""profiles/<round>_* from what scripts/profile_round5.sh (rounds 3, 4: profile_round3.sh, profile_round4.sh) left under gpurun_out/<tag>_*:   python3 scripts/collect_profiles.py <tag> <round, e.g. r04>
Copies the bench lines and the kernel-statistics summaries, and builds the PMC traffic and SQ counter summaries (scripts/make_pmc_profile.py's units)."""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
prof = os.path.join(G, tag + "_prof")


def last_json_line(path):
    return [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1]


open(os.path.join(P, rnd + "_bench.json"), "w").write(last_json_line(os.path.join(G, tag + "_bench.json")) + "\n")
open(os.path.join(P, rnd + "_bench_under_rocprof.json"), "w").write(last_json_line(os.path.join(prof, "bench_under_prof.json")) + "\n")
# (the two-ranks-on-one-GPU statistics are copied only on request, argv[3] == "halo": a launch there waits for the other rank's host thread, and a run in
# which that thread was late -- seconds in one maximum -- says nothing about the kernels)
halo = (("halo", "two_ranks_one_gpu_inkernel_halo_kernel_stats"), ("halo_resident", "two_ranks_one_gpu_resident_halo_kernel_stats")) if sys.argv[3:4] == ["halo"] else ()
n2 = os.path.join(G, tag + "_bench_n2_one_gpu.json")
if os.path.exists(n2) and os.path.getsize(n2) > 0:
    open(os.path.join(P, rnd + "_bench_n2_one_gpu.json"), "w").write(last_json_line(n2) + "\n")
for sub, name in (("stats", "bench_kernel_stats"), ("stats182", "resident_182k_kernel_stats"), ("stats367", "resident_367k_kernel_stats"), ("half", "rank_of_two_loopback_kernel_stats"), ("eighth", "rank_of_eight_loopback_kernel_stats")) + halo:
    f = glob.glob(os.path.join(prof, sub, "**", "*kernel_stats.csv"), recursive=True)
    if f: shutil.copy(f[0], os.path.join(P, f"{rnd}_{name}.csv"))
stats = os.path.join(P, rnd + "_bench_kernel_stats.csv")
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_pmc_profile.py"), os.path.join(prof, "pmc"), stats, os.path.join(P, rnd + "_pmc_traffic.json"),
                       "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only) over scripts/run_steps.py --mesh 2km --steps 1 --graph 0 "
                       f"(scripts/profile_round5.sh; round 4: profile_round4.sh); avg_ns from profiles/{rnd}_bench_kernel_stats.csv"], stdout=subprocess.DEVNULL)
if os.path.isdir(os.path.join(prof, "pmc_half", "FETCH_SIZE")) and os.path.exists(os.path.join(P, rnd + "_rank_of_two_loopback_kernel_stats.csv")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_pmc_profile.py"), os.path.join(prof, "pmc_half"), os.path.join(P, rnd + "_rank_of_two_loopback_kernel_stats.csv"),
                           os.path.join(P, rnd + "_rank_of_two_loopback_pmc_traffic.json"),
                           "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, counters only) over scripts/run_steps.py --mesh 2km --nparts 2 --rank 0 --loopback "
                           f"--opt fused=3 --opt halo_fused=1 --steps 1 --graph 0 (rank 0's half of the 2 km mesh alone on the device, its mailboxes connected to themselves); avg_ns from "
                           f"profiles/{rnd}_rank_of_two_loopback_kernel_stats.csv"], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "resource_usage.py")], stdout=open(os.path.join(P, rnd + "_resource_usage.json"), "w"))
res = json.load(open(os.path.join(P, rnd + "_resource_usage.json")))["kernels"]
res = {k.split("(")[0].replace("void ", ""): v for k, v in res.items()}
avg_ns = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) for r in csv.DictReader(open(stats))}
f = glob.glob(os.path.join(prof, "sq", "**", "*counter_collection.csv"), recursive=True)
if f:
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    out = {}
    for k in acc:
        if k.startswith("__amd"): continue
        e = {c: acc[k][c] / n[k][c] for c in acc[k]}
        e["launches_sampled"] = max(n[k].values())
        if k in avg_ns:
            e["avg_ns"] = avg_ns[k]
            e["valu_busy_estimate"] = round(e.get("SQ_INSTS_VALU", 0.) * 4 / (1024 * avg_ns[k] * 2.4), 3)
        rk = res.get(k) or res.get("void " + k)
        if rk: e["registers"] = {q: rk.get(q) for q in ("vgprs", "vgprs_spill", "totalsgprs", "sgprs_spill", "scratchsize", "occupancy")}
        out[k] = e
    json.dump({"source": "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, counters only) over scripts/run_steps.py --mesh 2km --steps 1 "
                         f"--graph 0 (scripts/profile_round5.sh; round 4: profile_round4.sh); averages per launch; avg_ns from profiles/{rnd}_bench_kernel_stats.csv; valu_busy_estimate = SQ_INSTS_VALU x 4 "
                         "cycles / (1024 SIMDs x avg_ns x 2.4 GHz) -- an upper bound (not every VALU instruction is fp64); registers from the code objects "
                         f"(profiles/{rnd}_resource_usage.json)", "kernels": out}, open(os.path.join(P, rnd + "_sq_counters.json"), "w"), indent=1)
print("collected into profiles/:", sorted(x for x in os.listdir(P) if x.startswith(rnd)))
