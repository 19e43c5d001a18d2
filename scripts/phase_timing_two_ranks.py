import json, os, sys, tempfile, pathlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_multirank as T
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
KIND = sys.argv[2] if len(sys.argv) > 2 else "h16000"
os.environ["NXS_PHASE_DUMP"] = "/tmp"
os.environ["NXS_DYN_LIBRARY"] = os.path.join(ROOT, "nextsim_amd", "csrc", "libnxsdyn_phase.so")
with tempfile.TemporaryDirectory() as d:
    reps = T._run(2, KIND, 8, pathlib.Path(d), "ipc", over={"options": {"fused": 4, "patch_nodes": 180, "resident_overlap": 0}}, ranks_per_proc=2)
print([r["ok"] for r in reps], reps[0]["timing"]["substeps_ms"])
t = np.load("/tmp/phase0.npy").reshape(8192, 8)[:, :6]
t = t[t[:, 0] > 0]
d = np.diff(t, axis=1) * 10e-3
names = ("element phase (to barrier 1)", "node phase + stores issued", "stores drained + barrier", "publish + wait for neighbours (ranks and patches) + barrier", "halo loads + barrier")
for lab, sel in (("boundary patches (first %d)" % nb, slice(0, nb)), ("interior patches", slice(nb + 40, None))):
    print(lab, d[sel].shape[0])
    for nm, col in zip(names, d[sel].T):
        print(f"  {nm:62s} mean {col.mean():6.2f} us   p10 {np.percentile(col, 10):6.2f}   p90 {np.percentile(col, 90):6.2f}")
    print(f"  one sub-step {(t[sel, 5] - t[sel, 0]).mean() * 10e-3:6.2f} us")
