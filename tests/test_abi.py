"""The C-ABI library: loads, exports every symbol include/nxs_dyn.h declares, ctypes layouts match the
C structs, and -- on a box without a GPU -- refuses to work instead of falling back to the CPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

from nextsim_amd import _abi, dynamics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nxs_dyn.h")


def test_every_declared_symbol_is_exported():
    text = open(HEADER).read()
    declared = set(re.findall(r"NXS_API\s+(?:const\s+char\s*\*|int)\s*(nxs_\w+)\s*\(", text))
    assert declared and declared == set(dynamics.EXPORTS)
    L = dynamics.load_library()
    for name in declared:
        assert hasattr(L, name), name
    assert L.nxs_dyn_abi_version() == 2
    itext = open(os.path.join(ROOT, "include", "nxs_interp.h")).read()
    ideclared = set(re.findall(r"NXS_INTERP_API\s+(?:const\s+char\s*\*|int)\s*(nxs_\w+)\s*\(", itext))
    assert ideclared == set(dynamics.INTERP_EXPORTS)
    for name in ideclared:
        assert hasattr(L, name), name


def test_library_does_not_depend_on_the_oracle_or_torch():
    out = subprocess.check_output(["readelf", "-d", dynamics._LIB_PATH], text=True)
    needed = re.findall(r"NEEDED.*\[(.*)\]", out)
    assert not any("oracle" in n or "torch" in n or "c10" in n for n in needed), needed
    assert any("amdhip64" in n for n in needed)


def test_ctypes_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    structs = {"nxs_dyn_params": _abi.Params, "nxs_dyn_mesh": _abi.Mesh, "nxs_dyn_halo": _abi.Halo,
               "nxs_dyn_state": _abi.State, "nxs_dyn_forcing": _abi.Forcing, "nxs_dyn_diag": _abi.Diag,
               "nxs_dyn_timing": _abi.Timing, "nxs_dyn_traffic": _abi.Traffic}
    probes = [("nxs_dyn_params", "regrid_angle"), ("nxs_dyn_params", "young"), ("nxs_dyn_mesh", "nc_width"),
              ("nxs_dyn_mesh", "neumann_flags"), ("nxs_dyn_halo", "recv_index"), ("nxs_dyn_state", "drag_ui_young"),
              ("nxs_dyn_state", "conc_young"), ("nxs_dyn_timing", "substep_launches"), ("nxs_dyn_traffic", "move_ring_bytes"), ("nxs_dyn_traffic", "update_bytes")]
    body = "".join(f'printf("%zu\\n", sizeof({s}));' for s in structs)
    body += "".join(f'printf("%zu\\n", offsetof({s}, {f}));' for s, f in probes)
    src.write_text(f'#include <stdio.h>\n#include <stddef.h>\n#include "nxs_dyn.h"\nint main(void){{{body}return 0;}}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    expect = [C.sizeof(t) for t in structs.values()] + [getattr(structs[s], f).offset for s, f in probes]
    assert vals == expect


# (the default parameters, the physical constants and the enums are compared with the REFERENCE's values in tests/test_reference_constants.py)


def _has_gpu():
    try:
        out = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=30).stdout
        return "gfx9" in out
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present: the no-device path cannot be exercised")
def test_no_gpu_means_error_not_cpu_fallback():
    from nextsim_amd.forcing import default_params
    with pytest.raises(dynamics.NxsError) as e:
        dynamics.FiniteElementDynamics(default_params())
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "nextsim_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn)).read()
                assert "pyoracle" not in text and "dyn_ref" not in text and "liboracle" not in text, fn


def test_every_entry_point_is_a_function_try_block():
    """'Never throws across the ABI' (include/nxs_dyn.h): every extern "C" definition of the library with a body of its own is
    a function-try-block whose handler maps the exception to a status code (csrc/nxs_guard.hpp).  Checked on the source text, so that
    an entry point added later cannot forget it."""
    csrc = os.path.join(ROOT, "nextsim_amd", "csrc")
    sig = re.compile(r'^(?:extern "C" )?(?:int|void) (nxs_\w+)\(')
    guarded, total = 0, 0
    for fn in ("nxs_dyn.hip", "nxs_interp.hip", "nxs_krylov.hip", "nxs_io.cpp", "nxs_mesh.cpp"):
        lines = open(os.path.join(csrc, fn)).read().split("\n")
        for i, line in enumerate(lines):
            m = sig.match(line)
            if not m:
                continue
            j = i
            while not lines[j].split("//")[0].rstrip().endswith(("{", "}", ";")):
                j += 1
            head = lines[j].split("//")[0].rstrip()
            if not head.endswith("{"):
                continue                       # a declaration, or a one-line body that only returns a member
            total += 1
            assert head.endswith("try {"), f"{fn}:{j + 1}: {m.group(1)} is not a function-try-block"
            k = j + 1
            while not lines[k].startswith("}"):
                k += 1
            assert "catch (...)" in lines[k] and m.group(1) in lines[k], f"{fn}:{k + 1}: handler of {m.group(1)}"
            guarded += 1
    assert guarded == total and total >= 65, (guarded, total)


def test_allocation_failure_inside_the_library_is_a_status_code(tmp_path):
    """A std::bad_alloc inside an entry point comes back as NXS_ERR_NOMEM (-6), not as std::terminate: with the address space capped just above
    what the process already uses, nxs_mesh_connectivity is asked for the tables of 60 M nodes (its first vector alone needs 240 MB)."""
    code = r'''
import ctypes as C, resource, sys
import numpy as np
sys.path.insert(0, %r)
from nextsim_amd import dynamics
L = dynamics.load_library()
idx = np.array([1, 2, 3], np.int32)
w1, w2 = C.c_int32(), C.c_int32()
vm = int(open("/proc/self/statm").read().split()[0]) * resource.getpagesize()
resource.setrlimit(resource.RLIMIT_AS, (vm + (128 << 20), resource.RLIM_INFINITY))
rc = L.nxs_mesh_connectivity(idx.ctypes.data_as(C.POINTER(C.c_int32)), 60_000_000, 1, C.byref(w1), None, C.byref(w2), None)
print("rc", rc)
''' % ROOT
    r = subprocess.run([os.sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rc -6" in r.stdout, (r.stdout, r.stderr[-2000:])
