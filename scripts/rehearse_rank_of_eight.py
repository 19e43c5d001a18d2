"""What a CU does on one rank of eight, rehearsed on ONE GPU: the 182 k-triangle mesh (1/8 of the 2 km mesh) over TWO ranks hosted by one process
(two threads, two handles, in-process mailboxes), both cut into the 180-node patches a rank of eight has, and the exchange between the ranks runs inside
the resident launch (minus xGMI: both ranks share the device).  Ranks that share a device get 70 % of its resident workgroup slots between them
(csrc/nxs_resident_registry.hpp): the default mesh h19000 (124 k triangles) gives 2 x 175 workgroups = 68 %; h16000 (173 k, 2 x 245 = 96 %, every CU holding two
resident patches as on the real machine) lost steps in round 3 and is now refused up front -- the second rank then runs one kernel per sub-step -- unless
NXS_RESIDENT_SHARED_LIMIT=100 is set knowingly.  Prints each rank's per-step timing for the resident launch (plain and with the interior elements under the exchange) and checks
the bits against the separate kernels and the oracle.        python3 scripts/rehearse_rank_of_eight.py [steps]
NXS_PN=360: the same partition as 2 x 123 patches of 360 nodes, one 512-thread workgroup per CU with several elements per thread (k_substep_resident_big)."""
import json, os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_multirank as T
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
KIND = sys.argv[2] if len(sys.argv) > 2 else "h19000"   # 124 k triangles: two ranks of ~175 patches of 180 nodes = 68 % of the 512 slots
for overlap in (0, 1):
    with tempfile.TemporaryDirectory() as d:
        reps = T._run(2, KIND, steps, pathlib.Path(d), "ipc", over={"options": {"fused": 4, "patch_nodes": int(os.environ.get("NXS_PN", "180")), "band_patch_nodes": int(os.environ.get("NXS_BAND", "-1")), "resident_overlap": overlap, "resident_wide": int(os.environ.get("NXS_WIDE", "0"))}}, ranks_per_proc=int(os.environ.get("NXS_RPP", "2")))
    for r in reps:
        tm = r.get("timing", {})
        if not r["ok"]: print("   ", {k: v for k, v in r.items() if k not in ("timing",)})
        print(f"overlap {overlap} rank {r['rank']}: ok {r['ok']}, launches per step {r.get('launches_fused')}, bits equal to the separate kernels {r.get('fused_equals_separate')}, "
              f"substeps {tm.get('substeps_ms', 0):.3f} ms, smoother {tm.get('smoother_ms', 0):.3f} ms, step {tm.get('total_ms', 0):.3f} ms; worst error vs the 2-rank oracle "
              f"{max(r.get('errs', {'-': 0}).values()):.2e}", flush=True)
