"""The registry of resident grids (nextsim_amd/csrc/nxs_resident_registry.hpp, the text libnxsdyn.so includes) driven without a device by
tests/native/registry_host.cpp: the claim arithmetic, and -- across PROCESSES, through the POSIX shared-memory table -- a second process'
grid refused up front, a claim within the shared limit admitted, a killed process' claim reclaimed."""
import os
import signal
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("reg") / "registry_host"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall",
                           "-I", os.path.join(ROOT, "nextsim_amd", "csrc"), os.path.join(ROOT, "tests", "native", "registry_host.cpp"), "-o", str(out), "-lrt", "-pthread"])
    return str(out)


def _env():
    prefix = f"nxs_test_{os.getpid()}_"
    return dict(os.environ, NXS_RESIDENT_SHM_PREFIX=prefix, ASAN_OPTIONS="detect_leaks=0"), prefix


def _cleanup(prefix):
    for fn in os.listdir("/dev/shm") if os.path.isdir("/dev/shm") else []:
        if fn.startswith(prefix):
            os.unlink(os.path.join("/dev/shm", fn))


def test_claim_arithmetic(exe):
    env, prefix = _env()
    try:
        out = subprocess.run([exe, "selftest", "arith"], env=env, capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and "selftest ok" in out.stdout, out.stderr + out.stdout
    finally:
        _cleanup(prefix)


def test_a_second_process_is_refused_up_front_and_a_dead_one_is_forgotten(exe):
    env, prefix = _env()
    holder = None
    try:
        holder = subprocess.Popen([exe, "hold", "dev0", "500", "512", "0"], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        first = holder.stdout.readline()
        assert first.startswith("claimed 1"), first
        if "shared" not in first:
            pytest.skip("no POSIX shared memory here: the registry is per process")

        def attempt(wg, slots, multi, ord_=()):
            r = subprocess.run([exe, "try", "dev0", str(wg), str(slots), str(multi)] + [str(v) for v in ord_], env=env, capture_output=True, text=True, timeout=60)
            assert r.returncode == 0, r.stderr
            return r.stdout
        # 97.7 % held by another process: a second resident grid that does not fit beside it is refused before it is built -- by arithmetic, not by a constant:
        # the holder is a single-rank handle whose kernels never wait, so what is left of the device is the claimant's
        out = attempt(100, 512, 0)
        assert out.startswith("claimed 0") and "of another process" in out and "2 handles" in out, out
        out = attempt(13, 512, 1)                                # 97.7 + 2.5
        assert out.startswith("claimed 0"), out
        out = attempt(5, 512, 1)                                 # 97.7 + 1.0
        assert out.startswith("claimed 1"), out
        # the holder lets go (its process ends in an orderly way): the device is free
        holder.stdin.write("\n"); holder.stdin.flush(); holder.wait(timeout=30)
        out = attempt(512, 512, 0)
        assert out.startswith("claimed 1") and "1 handles" in out, out
        # two processes side by side: their claims AND the blocks of each other's ordinary kernels that may wait (172 of 2 048: 8.4 %) must fit the device
        holder = subprocess.Popen([exe, "hold", "dev0", "150", "512", "1", "172", "2048"], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        assert holder.stdout.readline().startswith("claimed 1")
        out = attempt(150, 512, 1, (172, 2048))
        assert out.startswith("claimed 1") and "0.586 claimed" in out, out
        out = attempt(318, 512, 1, (172, 2048))                  # 29.3 + 62.1 + 8.4 = 99.8 %
        assert out.startswith("claimed 1"), out
        out = attempt(330, 512, 1, (172, 2048))                  # 29.3 + 64.5 + 8.4 = 102 %
        assert out.startswith("claimed 0") and "may hold while they wait" in out and "of another process" in out, out
        # a holder that is KILLED never lets go: its entry is dropped by the next process that looks (pid + start time)
        holder.send_signal(signal.SIGKILL); holder.wait(timeout=30)
        out = attempt(512, 512, 1)
        assert out.startswith("claimed 1") and "1 handles" in out, out
    finally:
        if holder and holder.poll() is None:
            holder.kill()
        _cleanup(prefix)
