"""Registers, spills, scratch and LDS of every kernel of libnxsdyn.so, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (make -C
nextsim_amd/csrc resource-usage), as JSON.   python3 scripts/resource_usage.py > profiles/rNN_resource_usage.json"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "nextsim_amd", "csrc")
out = subprocess.run(["make", "-C", csrc, "resource-usage"], capture_output=True, text=True).stderr
kernels, cur = {}, None
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
        name = re.sub(r"\(Dev.*", "", name).replace("void ", "")
        cur = kernels.setdefault(name, {})
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        key = m.group(1).strip().lower().replace(" ", "_")
        cur[key] = int(m.group(2))
json.dump({"source": "hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (make resource-usage), nxs_dyn.hip", "kernels": kernels}, sys.stdout, indent=1, sort_keys=True)
