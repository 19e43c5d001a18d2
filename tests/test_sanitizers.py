"""The reference's DEBUG build is an AddressSanitizer build (model/Makefile:21-25).  Same idea
for everything here that runs on the host: the oracle (oracle/dyn_ref.c), the host-side sources of the product
library (nxs_mesh.cpp, nxs_io.cpp), the host build of the remapping kernel's per-triangle functions, and the host-only mesh
preparation behind nxs_dyn_set_mesh (nxs_patchcut.hpp: patch cutter, Hilbert re-cut, exchange and resident tables; nxs_hull.inl:
the convex completion -- through tests/native/patchcut_host.cpp, which also checks every table it builds) are compiled
with -fsanitize=address,undefined and driven through their edge cases (tests/sanitize_worker.py).  GPU code cannot
be sanitized on this pool; its memory safety rests on the host-side shape checks in front of every launch."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1", "-fPIC", "-shared"]


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "nextsim_amd", "csrc")
    oracle, host, remap, cut = (str(tmp_path / n) for n in ("liboracle_san.so", "libhost_san.so", "libremap_san.so", "libpatchcut_san.so"))
    subprocess.check_call(["gcc", "-std=c11", "-ffp-contract=off"] + SAN + ["-o", oracle, os.path.join(ROOT, "oracle", "dyn_ref.c"), "-lm"])
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include")] + SAN +
                          ["-o", host, os.path.join(csrc, "nxs_mesh.cpp"), os.path.join(csrc, "nxs_io.cpp")])
    subprocess.check_call(["g++", "-std=c++14", "-ffp-contract=off"] + SAN + ["-o", remap, os.path.join(ROOT, "oracle", "remap_host.cpp")])
    subprocess.check_call(["g++", "-std=c++17", "-D_GLIBCXX_ASSERTIONS"] + SAN + ["-o", cut, os.path.join(ROOT, "tests", "native", "patchcut_host.cpp")])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    ubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               NXS_ORACLE_LIBRARY=oracle)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize_worker.py"), host, remap, cut], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "sanitize worker ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
