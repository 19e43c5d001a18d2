// registry_host.cpp -- drives nextsim_amd/csrc/nxs_resident_registry.hpp (the very text libnxsdyn.so includes) without a device, from one or several
// processes (tests/test_resident_registry.py).
//   registry_host selftest <key>                      the claim arithmetic inside one process; prints "selftest ok"
//   registry_host hold <key> <wg> <slots> <multi> [ord_blocks ord_slots]    registers a handle (and its waiting ordinary grid), claims, prints "claimed 0|1 <why>",
//                                                     then waits for a line on stdin and lets go
//   registry_host try  <key> <wg> <slots> <multi> [ord_blocks ord_slots]    registers, claims, prints "claimed 0|1 <why>" and the device's totals, lets go
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <sys/file.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "nxs_resident_registry.hpp"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "registry_host: check failed at line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1], key = argv[2];
    nxs_reg::DeviceTable &T = nxs_reg::table_for(key);
    if (mode == "selftest") {
        std::string why;
        const uint64_t a = 0x1000, b = 0x2000, c = 0x3000;
        CHECK(!T.claim(a, 10, 512, false));                       // not registered
        T.add(a);
        CHECK(T.handles() == 1);
        CHECK(T.claim(a, 512, 512, false) && T.claimed() == 1.0); // the only handle: the whole device
        CHECK(!T.claim(a, 513, 512, false, &why) && !why.empty() && T.claimed() == 0.);  // a refused claim leaves nothing behind
        CHECK(T.claim(a, 511, 512, true));                        // several ranks, but alone on the device
        T.add(b);                                                 // an idle second handle that has registered no waiting grid (a single-rank handle)
        CHECK(T.claim(a, 511, 512, false) && T.claim(a, 511, 512, true));   // ... costs nothing: its kernels never wait for anybody
        // round 3's rehearsal, from the grids themselves: two ranks of 44 k own nodes on one device -- 172 blocks of k_smooth_halo each, of the 2 048 the device
        // holds (8.4 %) -- with 2 x 245 resident workgroups of 512 slots (47.9 % each) lost steps; with 2 x 174 (34 % each) 32 of 32 passes ran clean
        T.set_ordinary(a, 172, 2048); T.set_ordinary(b, 172, 2048);
        CHECK(T.waiting() > 0.167 && T.waiting() < 0.169);
        CHECK(T.claim(a, 245, 512, true));                        // 47.9 % + b's 8.4 % of waiting blocks
        CHECK(!T.claim(b, 245, 512, true, &why));                 // 47.9 + 47.9 + a's 8.4 = 104 %: refused up front
        CHECK(why.find("may hold while they wait") != std::string::npos);
        CHECK(T.claim(b, 222, 512, true));                        // 47.9 + 43.4 + 8.4 = 99.6 %: the rule's edge
        CHECK(!T.claim(b, 226, 512, true));                       // 100.4 %
        CHECK(T.claim(a, 174, 512, true) && T.claim(b, 174, 512, true));   // the configuration that ran clean
        // a grid that lets go keeps its waiting grid registered; a larger waiting grid registered later shrinks what others may still claim
        T.release(b);
        T.set_ordinary(b, 1024, 2048);                            // 50 %
        CHECK(!T.claim(a, 300, 512, true) && T.claim(a, 256, 512, true));
        T.set_ordinary(b, 0, 0);
        T.add(c);
        CHECK(!T.claim(c, 2, 1, false));                          // more workgroups than the device holds of that build: never
        CHECK(T.claim(c, 10, 1024, false));
        // one claim per handle, held by one of its two grids: a release that names the other grid leaves it alone (release_resident wiped a live pair claim)
        CHECK(T.claim(c, 40, 512, true, nullptr, nxs_reg::KIND_PAIR));
        const double before = T.claimed();
        T.release(c, nxs_reg::KIND_RESIDENT);
        CHECK(T.claimed() == before);
        T.release(c, nxs_reg::KIND_PAIR);
        CHECK(T.claimed() < before);
        T.release(a);
        CHECK(T.claim(b, 300, 512, true));                        // a's share is free again
        // entries this code did not write are ignored and cleared; an entry of ANOTHER PID namespace is not judged by its pid -- it expires
        nxs_reg::Entry bad{}; bad.pid = 1; bad.wg = -5; bad.slots = 3; bad.pidns = nxs_reg::my_pid_namespace();
        T.plant(bad);
        CHECK(T.handles() == 3);                                  // (swept on the way in)
        nxs_reg::Entry foreign{}; foreign.pid = 999999; foreign.start = 1; foreign.handle = 0x77; foreign.wg = 400; foreign.slots = 512;
        foreign.pidns = nxs_reg::my_pid_namespace() + 1; foreign.stamp = nxs_reg::boot_seconds();
        T.plant(foreign);
        CHECK(T.handles() == 4);                                  // pid 999999 does not exist HERE, the entry stays: its owner lives in another namespace
        T.release(b);
        CHECK(!T.claim(b, 300, 512, true, &why) && why.find("of another process") != std::string::npos);
        foreign.stamp = nxs_reg::boot_seconds() > nxs_reg::STALE_SECONDS + 5 ? nxs_reg::boot_seconds() - nxs_reg::STALE_SECONDS - 5 : 0; foreign.handle = 0x78;
        T.plant(foreign);                                         // the same, untouched for more than STALE_SECONDS: dropped by the next sweep
        CHECK(T.handles() == 4);
        nxs_reg::Entry dead{}; dead.pid = 999998; dead.start = 1; dead.handle = 0x79; dead.wg = 100; dead.slots = 512; dead.pidns = nxs_reg::my_pid_namespace(); dead.stamp = nxs_reg::boot_seconds();
        T.plant(dead);                                            // this namespace, no such process: dropped at once
        CHECK(T.handles() == 4);
        T.remove(b); T.remove(c);
        CHECK(T.handles() == 2 && T.claimed() > 0.78);            // a + the live foreign entry
        T.remove(a);
        if (T.shared()) {   // a fork()ed child is excluded by the lock like any other process: it blocks while the parent holds it (with the parent's description it would walk through)
            CHECK(flock(T.lock_fd(), LOCK_EX) == 0);
            const pid_t child = fork();
            if (child == 0) {
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                (void)T.handles();
                clock_gettime(CLOCK_MONOTONIC, &t1);
                _exit((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) >= 0.2 ? 0 : 1);
            }
            CHECK(child > 0);
            usleep(400000);
            CHECK(flock(T.lock_fd(), LOCK_UN) == 0);
            int status = 0;
            CHECK(waitpid(child, &status, 0) == child && WIFEXITED(status) && WEXITSTATUS(status) == 0);
        }
        printf("selftest ok (%s)\n", T.shared() ? "shared memory" : "process-local");
        return 0;
    }
    if (argc < 6) return 2;
    const int wg = atoi(argv[3]), slots = atoi(argv[4]);
    const bool multi = atoi(argv[5]) != 0;
    const uint64_t me = 0xabc000;
    T.add(me);
    if (argc >= 8) T.set_ordinary(me, atoi(argv[6]), atoi(argv[7]));
    std::string why;
    const bool ok = T.claim(me, wg, slots, multi, &why);
    printf("claimed %d %s | device: %.3f claimed, %d handles, %s\n", ok ? 1 : 0, why.c_str(), T.claimed(), T.handles(), T.shared() ? "shared" : "local");
    fflush(stdout);
    if (mode == "hold") { char line[16]; if (!fgets(line, sizeof line, stdin)) {} }
    T.remove(me);
    return 0;
}
