// nextsim_mpi.cpp -- the multi-rank host in the reference's own terms: one MPI rank per GPU (the reference uses Boost.MPI over
// the same MPI), the mesh partition and halo lists each rank got from distributedMeshProcessing / initUpdateGhosts, and
// libnxsdyn.so under step().  The halo transport is the device-direct one (peer mailboxes mapped with hipIpc, exchange
// inside the sub-step kernel); MPI only carries one record per rank once, at set-up -- or, with
// "host" as the last argument, the literal send / recv of FE.cpp:13981-13985 through MPI.  The halo lists go in VERBATIM, also
// where a ragged partition sends to a rank it receives nothing from.  Host code is plain C++14 + MPI (no hipcc, no Python):
//     mpicxx -std=c++14 -O2 -Iinclude examples/nextsim_mpi.cpp -Lnextsim_amd/csrc -lnxsdyn -Wl,-rpath,$PWD/nextsim_amd/csrc -o nextsim_mpi
//     mpiexec -n 2 ./nextsim_mpi case_%d.bin 2 out_%d.bin [gpus_per_node [ipc|host]]     (%d = rank; case files from nextsim_amd/casefile.py)
#include <mpi.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "nxs_dyn.hpp"

struct CaseFile {
    std::map<std::string, std::vector<double>> dbl;
    std::map<std::string, std::vector<int32_t>> i32;
    std::map<std::string, std::vector<uint8_t>> u8;
    nxs_dyn_params params{};
};

static CaseFile read_case(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    CaseFile c;
    char magic[8];
    uint64_t psize = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "NXSCASE1", 8) || fread(&psize, 8, 1, f) != 1 || psize != sizeof(nxs_dyn_params) ||
        fread(&c.params, sizeof c.params, 1, f) != 1)
        throw std::runtime_error("not a case file: " + path);
    for (;;) {
        uint32_t nlen = 0, kind = 0;
        uint64_t count = 0;
        if (fread(&nlen, 4, 1, f) != 1) break;
        std::string name(nlen, ' ');
        if (fread(&name[0], 1, nlen, f) != nlen || fread(&kind, 4, 1, f) != 1 || fread(&count, 8, 1, f) != 1) throw std::runtime_error("short read");
        size_t got = 0;
        if (kind == 0) { auto &v = c.dbl[name]; v.resize(count); got = fread(v.data(), 8, count, f); }
        else if (kind == 1) { auto &v = c.i32[name]; v.resize(count); got = fread(v.data(), 4, count, f); }
        else { auto &v = c.u8[name]; v.resize(count); got = fread(v.data(), 1, count, f); }
        if (got != count) throw std::runtime_error("short read");
    }
    fclose(f);
    return c;
}

static std::string with_rank(const char *pattern, int rank) {
    char buf[1024];
    snprintf(buf, sizeof buf, pattern, rank);
    return buf;
}

int main(int argc, char **argv) {
    MPI_Init(&argc, &argv);
    int rank = 0, nranks = 1, rc = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &nranks);
    try {
        if (argc < 4) throw std::runtime_error("usage: nextsim_mpi case_%d.bin nsteps out_%d.bin [gpus_per_node [ipc|host]]");
        CaseFile c = read_case(with_rank(argv[1], rank));
        const int nsteps = std::atoi(argv[2]), gpus = argc > 4 ? std::atoi(argv[4]) : 1;
        const std::string transport = argc > 5 ? argv[5] : "ipc";
        auto &D = c.dbl; auto &I = c.i32; auto &B = c.u8;
        if (I["halo_rank"][0] != rank || I["halo_rank"][1] != nranks) throw std::runtime_error("case file is for another rank / communicator size");

        nxs_dyn_mesh m{};
        m.num_nodes = I["sizes"][0]; m.num_elements = I["sizes"][1]; m.local_ndof = I["sizes"][2]; m.local_nelements = I["sizes"][3];
        m.indices = I["indices"].data(); m.ghost_nodes = B["ghost_nodes"].data();
        m.coord_x = D["coord_x"].data(); m.coord_y = D["coord_y"].data(); m.lat = D["lat"].data();
        m.mask_dirichlet = B["mask_dirichlet"].data();
        m.num_neumann_flags = (int32_t)I["neumann_flags"].size(); m.neumann_flags = I["neumann_flags"].data();
        nxs::FiniteElementDynamics FE(c.params, rank % gpus);
        FE.setMesh(m);

        // ---- initUpdateGhosts' lists (FE.hpp:615-618) and the device-direct transport
        const int ns = (int)I["send_procs"].size(), nr = (int)I["recv_procs"].size();
        nxs_dyn_halo hl{rank, nranks, ns, nr, I["send_procs"].data(), I["send_offsets"].data(), I["send_index"].data(),
                        I["recv_procs"].data(), I["recv_offsets"].data(), I["recv_index"].data()};
        FE.setHalo(hl);
        struct HostHalo { CaseFile *c; int rank; } hh{&c, rank};
        if (transport == "host") {
            // the literal M_comm.send / M_comm.recv of FE.cpp:13981-13985 through the caller's communicator: the library hands over the packed segments
            // (per neighbour k of MY lists, 2 * n_k doubles at 2 * send_offsets[k]) and takes the received ones back
            FE.setHaloExchange([](void *ctx, const double *send, double *recv) -> int {
                auto *h = static_cast<HostHalo *>(ctx);
                auto &I = h->c->i32;
                const int ns = (int)I["send_procs"].size(), nr = (int)I["recv_procs"].size();
                std::vector<MPI_Request> rq;
                rq.reserve(ns + nr);
                for (int k = 0; k < nr; ++k) {
                    const int n = 2 * (I["recv_offsets"][k + 1] - I["recv_offsets"][k]);
                    rq.emplace_back();
                    MPI_Irecv(recv + 2 * (size_t)I["recv_offsets"][k], n, MPI_DOUBLE, I["recv_procs"][k], 77, MPI_COMM_WORLD, &rq.back());
                }
                for (int k = 0; k < ns; ++k) {
                    const int n = 2 * (I["send_offsets"][k + 1] - I["send_offsets"][k]);
                    rq.emplace_back();
                    MPI_Isend(const_cast<double *>(send) + 2 * (size_t)I["send_offsets"][k], n, MPI_DOUBLE, I["send_procs"][k], 77, MPI_COMM_WORLD, &rq.back());
                }
                return MPI_Waitall((int)rq.size(), rq.data(), MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
            }, &hh);
        } else {
            // the device-direct mailboxes: every rank publishes ONE record (its mailbox handle + its receive lists as the library holds them -- the directions
            // nxs_dyn_set_halo added on a ragged partition included), MPI carries the records once, the library finds its segments itself
            int32_t nbytes = 0, stride = 0;
            if (nxs_dyn_ipc_record_bytes(FE.handle(), &nbytes)) throw std::runtime_error(nxs_dyn_last_error(FE.handle()));
            MPI_Allreduce(&nbytes, &stride, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
            std::vector<char> rec(stride), recs((size_t)stride * nranks);
            if (nxs_dyn_ipc_export_record(FE.handle(), rec.data(), stride)) throw std::runtime_error(nxs_dyn_last_error(FE.handle()));
            MPI_Allgather(rec.data(), stride, MPI_CHAR, recs.data(), stride, MPI_CHAR, MPI_COMM_WORLD);
            if (nxs_dyn_ipc_connect_records(FE.handle(), recs.data(), stride, nranks)) throw std::runtime_error(nxs_dyn_last_error(FE.handle()));
            int32_t err = 0, worst = 0;
            if (nxs_dyn_ipc_selftest(FE.handle(), 32, &err)) throw std::runtime_error(nxs_dyn_last_error(FE.handle()));
            MPI_Allreduce(&err, &worst, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
            if (worst) throw std::runtime_error("mailbox self-test failed on some rank");   // a production host falls back to RCCL / its own MPI here
        }

        nxs_dyn_state s{};
        s.VT = D["VT"].data(); s.UM = D["UM"].data(); s.UT = D["UT"].data();
        s.conc = D["conc"].data(); s.thick = D["thick"].data(); s.snow_thick = D["snow_thick"].data();
        s.damage = D["damage"].data(); s.ridge_ratio = D["ridge_ratio"].data();
        s.sigma[0] = D["sigma0"].data(); s.sigma[1] = D["sigma1"].data(); s.sigma[2] = D["sigma2"].data();
        s.conc_young = D["conc_young"].data(); s.h_young = D["h_young"].data(); s.hs_young = D["hs_young"].data();
        s.conc_myi = D["conc_myi"].data(); s.thick_myi = D["thick_myi"].data();
        s.cohesion = D["cohesion"].data(); s.time_relaxation_damage = D["time_relaxation_damage"].data();
        s.drag_ui = D["drag_ui"].data(); s.drag_ui_young = D["drag_ui_young"].data();
        nxs_dyn_forcing f{D["wind"].data(), D["ocean"].data(), D["ssh"].data(), D["element_depth"].data()};
        FE.putState(s);
        FE.setForcing(f);
        for (int pcpt = 0; pcpt < nsteps; ++pcpt) {
            int32_t crash = 0, any = 0;                                        // FE.cpp:7975 + the all_reduce of :14647
            nxs_dyn_check_fields_fast(FE.handle(), &crash);
            MPI_Allreduce(&crash, &any, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
            if (any) throw std::runtime_error("checkFieldsFast: a field is out of range on some rank");
            FE.step();
        }
        FE.synchronize();
        FE.getState(s);
        FILE *o = fopen(with_rank(argv[3], rank).c_str(), "wb");
        if (!o) throw std::runtime_error("cannot write the output");
        for (const char *k : {"VT", "UM", "UT", "sigma0", "sigma1", "sigma2", "damage", "conc", "thick"}) fwrite(D[k].data(), 8, D[k].size(), o);
        fclose(o);
        if (rank == 0) std::cout << "nextsim_mpi: " << nranks << " ranks, " << nsteps << " steps done\n";
    } catch (const std::exception &ex) {
        std::cerr << "nextsim_mpi[" << rank << "]: " << ex.what() << "\n";
        rc = 1;
        MPI_Abort(MPI_COMM_WORLD, 1);
    }
    MPI_Finalize();
    return rc;
}
