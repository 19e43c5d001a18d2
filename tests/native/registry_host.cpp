// registry_host.cpp -- drives nextsim_amd/csrc/nxs_resident_registry.hpp (the very text libnxsdyn.so includes) without a device, from one or several
// processes (tests/test_resident_registry.py).
//   registry_host selftest <key>                      the claim arithmetic inside one process; prints "selftest ok"
//   registry_host hold <key> <wg> <slots> <multi>     registers a handle, claims, prints "claimed 0|1 <why>", then waits for a line on stdin and lets go
//   registry_host try  <key> <wg> <slots> <multi>     registers, claims, prints "claimed 0|1 <why>" and the device's totals, lets go
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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
        T.add(b);                                                 // an idle second handle
        CHECK(T.claim(a, 511, 512, false));                       // a single-rank claimant keeps the whole device beside an idle handle ...
        CHECK(!T.claim(a, 511, 512, true, &why));                 // ... a several-rank one does not: its neighbours' kernels need room
        CHECK(T.claim(a, 245, 512, true));                        // 48 %
        CHECK(!T.claim(b, 245, 512, true, &why));                 // 96 % together: the configuration that lost steps in round 3 is refused up front
        CHECK(why.find("headroom") != std::string::npos);
        CHECK(T.claim(b, 100, 512, true));                        // 67 % together
        T.add(c);
        CHECK(!T.claim(c, 1, 1, false));                          // a build of which the device holds ONE workgroup: 100 % on its own
        CHECK(T.claim(c, 10, 1024, false));
        T.release(a);
        CHECK(T.claim(b, 300, 512, true));                        // a's share is free again
        T.remove(b); T.remove(c);
        CHECK(T.handles() == 1 && T.claimed() == 0.);
        CHECK(T.claim(a, 512, 512, true));
        T.remove(a);
        CHECK(T.handles() == 0);
        printf("selftest ok (%s)\n", T.shared() ? "shared memory" : "process-local");
        return 0;
    }
    if (argc < 6) return 2;
    const int wg = atoi(argv[3]), slots = atoi(argv[4]);
    const bool multi = atoi(argv[5]) != 0;
    const uint64_t me = 0xabc000;
    T.add(me);
    std::string why;
    const bool ok = T.claim(me, wg, slots, multi, &why);
    printf("claimed %d %s | device: %.3f claimed, %d handles, %s\n", ok ? 1 : 0, why.c_str(), T.claimed(), T.handles(), T.shared() ? "shared" : "local");
    fflush(stdout);
    if (mode == "hold") { char line[16]; if (!fgets(line, sizeof line, stdin)) {} }
    T.remove(me);
    return 0;
}
