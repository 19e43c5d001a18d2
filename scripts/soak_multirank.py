"""Soak of the several-rank paths on one shared GPU: 30 steps each, the resident launch (plain and with the interior elements under the exchange) and
the exchange inside one launch per sub-step against the separate push / pull kernels, bit for bit.   python3 scripts/soak_multirank.py"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_multirank as T
for world, kind, rpp, opts in ((8, "10km", 2, {"fused": 4}), (8, "10km", 2, {"fused": 4, "resident_overlap": 1}), (3, "40km", 1, {"fused": 4}), (2, "40km", 2, {"fused": 4, "resident_overlap": 1}), (4, "10km", 2, {"fused": 3}), (3, "40km", 1, {"fused": 4, "resident_wide": 1}), (2, "40km", 1, {"fused": 4, "resident_wide": 1, "resident_overlap": 1}),
                               (4, "10km", 2, {"fused": 4, "band_patch_nodes": 16}), (3, "40km", 1, {"fused": 4, "band_patch_nodes": 24, "patch_nodes": 128}), (2, "h19000", 2, {"fused": 4, "patch_nodes": 180})):
    with tempfile.TemporaryDirectory() as d:
        reps = T._run(world, kind, 30, pathlib.Path(d), "ipc", over={"options": opts, "dump": True}, ranks_per_proc=rpp)
    print(world, kind, opts, "ok", [r["ok"] for r in reps], "bits equal to separate", [r.get("fused_equals_separate") for r in reps], "launches", reps[0].get("launches_fused"), "crash", [r.get("crash") for r in reps], flush=True)
