import sys, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from nextsim_amd import mesh as M
gm = M.make_mesh("2km")
print(json.dumps(bench.aux_regrid(gm, with_cpu=True), indent=1))
