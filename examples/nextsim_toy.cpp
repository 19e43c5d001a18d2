// nextsim_toy.cpp -- BASELINE config 1 driven from C++: the reference's run()/step() skeleton
// (FE.cpp:8450-8502, 7963-8289) reduced to the dynamics path, on top of the C ABI.  Host code is plain
// C++14 built with g++ (no hipcc, no Python):
//     g++ -std=c++14 -O2 -Iinclude examples/nextsim_toy.cpp -Lnextsim_amd/csrc -lnxsdyn -Wl,-rpath,$PWD/nextsim_amd/csrc -o nextsim_toy
//     ./nextsim_toy case.bin 10 out_prefix
// case.bin is the flat dump written by nextsim_amd/casefile.py (mesh, parameters, state, forcing).
// Per step: checkFieldsFast, checkRegridding, step (explicitSolve + update); at the end the Exporter files
// field_final.bin/.dat + mesh_final.bin/.dat (FE.cpp:14111-14318) and the final state dump for the test.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "nxs_dyn.hpp"
#include "nxs_io.h"

struct CaseFile {
    std::map<std::string, std::vector<double>> dbl;
    std::map<std::string, std::vector<int32_t>> i32;
    std::map<std::string, std::vector<uint8_t>> u8;
    nxs_dyn_params params{};
};

static CaseFile read_case(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    CaseFile c;
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "NXSCASE1", 8)) throw std::runtime_error("not a case file");
    uint64_t psize = 0;
    if (fread(&psize, 8, 1, f) != 1 || psize != sizeof(nxs_dyn_params)) throw std::runtime_error("parameter block size mismatch");
    if (fread(&c.params, sizeof c.params, 1, f) != 1) throw std::runtime_error("short read");
    for (;;) {
        uint32_t nlen = 0, kind = 0; uint64_t count = 0;
        if (fread(&nlen, 4, 1, f) != 1) break;
        std::string name(nlen, ' ');
        if (fread(&name[0], 1, nlen, f) != nlen || fread(&kind, 4, 1, f) != 1 || fread(&count, 8, 1, f) != 1) throw std::runtime_error("short read");
        if (kind == 0) { auto &v = c.dbl[name]; v.resize(count); if (count && fread(v.data(), 8, count, f) != count) throw std::runtime_error("short read"); }
        else if (kind == 1) { auto &v = c.i32[name]; v.resize(count); if (count && fread(v.data(), 4, count, f) != count) throw std::runtime_error("short read"); }
        else { auto &v = c.u8[name]; v.resize(count); if (count && fread(v.data(), 1, count, f) != count) throw std::runtime_error("short read"); }
    }
    fclose(f);
    return c;
}

int main(int argc, char **argv) {
    if (argc < 4) { std::cerr << "usage: nextsim_toy case.bin nsteps out_prefix\n"; return 2; }
    try {
        CaseFile c = read_case(argv[1]);
        const int nsteps = std::atoi(argv[2]);
        const std::string prefix = argv[3];
        auto &D = c.dbl; auto &I = c.i32; auto &B = c.u8;

        nxs_dyn_mesh m{};
        m.num_nodes = I["sizes"][0]; m.num_elements = I["sizes"][1]; m.local_ndof = I["sizes"][2]; m.local_nelements = I["sizes"][3];
        m.indices = I["indices"].data(); m.ghost_nodes = B["ghost_nodes"].data();
        m.coord_x = D["coord_x"].data(); m.coord_y = D["coord_y"].data(); m.lat = D["lat"].data();
        m.mask_dirichlet = B["mask_dirichlet"].data();
        m.num_neumann_flags = (int32_t)I["neumann_flags"].size(); m.neumann_flags = I["neumann_flags"].data();
        // no bamg here: the library builds the two connectivity tables with bamg's ordering itself

        nxs::FiniteElementDynamics FE(c.params);
        FE.setMesh(m);
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
        FE.setForcing(f);   // constant forcing (nextsim.toy.cfg): one snapshot for the whole run

        for (int pcpt = 0; pcpt < nsteps; ++pcpt) {   // FiniteElement::run(), FE.cpp:8471-8494
            FE.checkFieldsFast();                     // FE.cpp:7975
            double ang = 0;
            if (FE.checkRegridding(&ang)) { std::cerr << "regrid needed at step " << pcpt << " (min angle " << ang << ")\n"; break; }  // FE.cpp:7988
            FE.step();                                // FE.cpp:8197-8214
        }
        FE.synchronize();
        FE.getState(s);

        // exportResults("final"), FE.cpp:14111-14318
        nxs_exporter *e = nullptr;
        std::vector<int32_t> ids(m.num_nodes);
        for (int i = 0; i < m.num_nodes; ++i) ids[i] = i + 1;
        if (nxs_exporter_open((prefix + "mesh_final.bin").c_str(), (prefix + "mesh_final.dat").c_str(), "double", &e)) throw std::runtime_error(nxs_io_last_error());
        nxs_exporter_write_mesh(e, m.coord_x, m.coord_y, ids.data(), m.num_nodes, m.indices, 3ll * m.num_elements);
        nxs_exporter_close(e);
        if (nxs_exporter_open((prefix + "field_final.bin").c_str(), (prefix + "field_final.dat").c_str(), "double", &e)) throw std::runtime_error(nxs_io_last_error());
        const double t[1] = {42291. + nsteps * c.params.dtime_step / 86400.};  // days since 1900-01-01 (2015-10-16, toy cfg)
        nxs_exporter_write_field(e, "Time", t, 1);
        for (const char *k : {"VT", "UM", "UT"}) nxs_exporter_write_field(e, (std::string("M_") + k).c_str(), D[k].data(), (int64_t)D[k].size());
        for (const char *k : {"conc", "thick", "damage", "ridge_ratio", "sigma0", "sigma1", "sigma2"})
            nxs_exporter_write_field(e, (std::string("M_") + k).c_str(), D[k].data(), (int64_t)D[k].size());
        nxs_exporter_close(e);
        std::cout << "nextsim_toy: " << nsteps << " steps on " << m.num_elements << " triangles done\n";
        return 0;
    } catch (const std::exception &ex) {
        std::cerr << "nextsim_toy: " << ex.what() << "\n";  // the reference lets it propagate to terminate()
        return 1;
    }
}
