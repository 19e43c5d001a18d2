// nxs_dyn.hpp -- header-only C++ convenience layer over the C ABI of nxs_dyn.h, with the reference's own
// method names and error behaviour: every failure throws std::runtime_error, as FiniteElement does
// (uncaught -> terminate, FE.cpp:14653).  Nothing here computes; it only forwards to libnxsdyn.so.
#ifndef NXS_DYN_HPP
#define NXS_DYN_HPP

#include <stdexcept>
#include <string>
#include <vector>

#include "nxs_dyn.h"

namespace nxs {

class FiniteElementDynamics {
public:
    explicit FiniteElementDynamics(const nxs_dyn_params &p, int device = 0) {
        if (nxs_dyn_create(&p, device, &h_) != 0) throw std::runtime_error(std::string("nxs_dyn_create: ") + nxs_dyn_last_error(nullptr));
    }
    ~FiniteElementDynamics() { nxs_dyn_destroy(h_); }
    FiniteElementDynamics(const FiniteElementDynamics &) = delete;
    FiniteElementDynamics &operator=(const FiniteElementDynamics &) = delete;

    void setMesh(const nxs_dyn_mesh &m) { check(nxs_dyn_set_mesh(h_, &m), "set_mesh"); }      // distributedMeshProcessing, FE.cpp:50-143
    void setHalo(const nxs_dyn_halo &hl) { check(nxs_dyn_set_halo(h_, &hl), "set_halo"); }    // initUpdateGhosts, FE.cpp:14003-14088
    void setHaloExchange(nxs_dyn_halo_fn fn, void *ctx) { check(nxs_dyn_set_halo_exchange_fn(h_, fn, ctx), "set_halo_exchange_fn"); }   // the caller's own M_comm.send / recv, FE.cpp:13981-13985
    void putState(const nxs_dyn_state &s) { check(nxs_dyn_put_state(h_, &s), "put_state"); }
    void getState(nxs_dyn_state &s) { check(nxs_dyn_get_state(h_, &s), "get_state"); }
    void setForcing(const nxs_dyn_forcing &f) { check(nxs_dyn_set_forcing(h_, &f), "set_forcing"); }
    // ExternalData's two snapshots resident + the per-step time coefficients (externaldata.cpp:360-401)
    void setForcingPair(const nxs_dyn_forcing &f0, const nxs_dyn_forcing &f1) { check(nxs_dyn_set_forcing_pair(h_, &f0, &f1), "set_forcing_pair"); }
    void setForcingTime(double fcoeff0, double fcoeff1, const double *factor = nullptr, const double *bias = nullptr) {
        check(nxs_dyn_set_forcing_time(h_, fcoeff0, fcoeff1, factor, bias), "set_forcing_time");
    }
    void setOption(const char *key, long long value) { check(nxs_dyn_set_option(h_, key, value), "set_option"); }
    void getDiag(nxs_dyn_diag &d) { check(nxs_dyn_get_diag(h_, &d), "get_diag"); }

    void explicitSolve() { check(nxs_dyn_explicit_solve(h_), "explicitSolve"); }               // FE.cpp:10182-10643
    void update() { check(nxs_dyn_update(h_), "update"); }                                     // FE.cpp:3919-4132
    void step() { check(nxs_dyn_step(h_), "step"); }                                           // FE.cpp:8197-8214
    void synchronize() { check(nxs_dyn_synchronize(h_), "synchronize"); }

    bool checkRegridding(double *min_angle = nullptr) {                                        // FE.cpp:8298-8309 (local part)
        double ang = 0; int32_t flip = 0, rg = 0;
        check(nxs_dyn_check_regridding(h_, &ang, &flip, &rg), "checkRegridding");
        if (min_angle) *min_angle = ang;
        return rg != 0;
    }
    void checkFieldsFast() {                                                                   // FE.cpp:14536-14655: throws on a bad field
        int32_t crash = 0;
        check(nxs_dyn_check_fields_fast(h_, &crash), "checkFieldsFast");
        if (crash) throw std::runtime_error("FiniteElement::checkFieldsFast: Check failed");
    }
    // FE.cpp:7860-7905: the element diagnostics of checkOutputs() / exportResults().  Host arrays of `d` that are not NULL are filled; the return value
    // is the DEVICE address of the same diagnostics as [Ne][NXS_ICE_DIAG_FIELDS] rows (what nxs_interp_mesh_to_grid_device samples for a Moorings record).
    const double *updateIceDiagnostics(nxs_dyn_ice_diag *d = nullptr) {
        const double *rows = nullptr;
        check(nxs_dyn_ice_diagnostics(h_, d, &rows), "updateIceDiagnostics");
        return rows;
    }
    nxs_dyn_handle *handle() { return h_; }

private:
    void check(int rc, const char *what) {
        if (rc != 0) throw std::runtime_error(std::string(what) + ": " + nxs_dyn_last_error(h_));
    }
    nxs_dyn_handle *h_ = nullptr;
};

}  // namespace nxs
#endif
