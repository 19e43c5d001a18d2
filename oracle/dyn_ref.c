/*
 * dyn_ref.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see dyn_ref.h for the rules and the
 * parity status).  Plain C11, single thread, fp64, reference loop order, flat arrays.
 *
 * Every function cites the reference lines it restates; "FE.cpp" is
 * /root/reference/model/finiteelement.cpp.  The reference quirks Q1..Q10 of SURVEY.md section 8(a)
 * are reproduced on purpose and marked where they occur.
 *
 * std::max(a,b) == (a<b)?b:a and std::min(a,b) == (b<a)?b:a -- NOT fmax/fmin (NaN handling and
 * signed zeros differ), hence the macros.
 */
#define _GNU_SOURCE
#define _POSIX_C_SOURCE 200809L  /* pthread barriers (ref_multirank_steps) under -std=c11 */
#include "dyn_ref.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define STD_MAX(a, b) (((a) < (b)) ? (b) : (a))
#define STD_MIN(a, b) (((b) < (a)) ? (b) : (a))

/* model/constants.hpp:56-87 */
static const double RHOI = 917.;
static const double RHOW = 1025.;
static const double RHOS = 330.;
static const double RHOA = 1.22;
static const double GRAVITY = 9.80616;
static const double OMEGA = 7.292e-5;
/* contrib/bamg/include/OppositeAngle.h:4 (what FE.cpp's PI resolves to) */
static const double PI_ = 3.141592653589793238462643383279502884197169399375105820974944592308;
static const double DAYS_IN_SEC = 86400.; /* model/finiteelement.hpp:549 */

/* ------------------------------------------------------------------------------------------ */

ref_work *ref_work_create(int32_t Nn, int32_t Ne) {
    ref_work *w = (ref_work *)calloc(1, sizeof(ref_work));
    if (!w) return NULL;
    w->Nn = Nn;
    w->Ne = Ne;
    size_t n = (size_t)(Nn > 0 ? Nn : 1), e = (size_t)(Ne > 0 ? Ne : 1);
    w->delta_x = (double *)calloc(e, sizeof(double));
    w->surface = (double *)calloc(e, sizeof(double));
    w->shape_coeff = (double *)calloc(6 * e, sizeof(double));
    w->B0T = (double *)calloc(18 * e, sizeof(double));
    w->element_mass = (double *)calloc(e, sizeof(double));
    w->rlmass_matrix = (double *)calloc(n, sizeof(double));
    w->node_mass = (double *)calloc(n, sizeof(double));
    w->C_bu = (double *)calloc(n, sizeof(double));
    w->grad_ssh = (double *)calloc(2 * n, sizeof(double));
    w->grad_terms = (double *)calloc(2 * n, sizeof(double));
    w->fcor = (double *)calloc(n, sizeof(double));
    w->VTM = (double *)calloc(2 * n, sizeof(double));
    w->tmp = (double *)calloc(2 * n, sizeof(double));
    w->D_tau_a = (double *)calloc(2 * n, sizeof(double));
    w->D_tau_w = (double *)calloc(2 * n, sizeof(double));
    w->D_del_ci_ridge_myi = (double *)calloc(e, sizeof(double));
    return w;
}

int ref_work_enable_trace(ref_work *w) {
    if (!w) return -1;
    free(w->trace);
    w->trace = (uint64_t *)calloc(4 * (size_t)(w->Ne > 0 ? w->Ne : 1), sizeof(uint64_t));
    return w->trace ? 0 : -1;
}

/* one sub-step of one element into the branch trace (see ref_work.trace) */
static void trace_branch(uint64_t *t, int code, double dcrit, double conc) {
    t[0] = t[0] * 0x9E3779B97F4A7C15ull + (uint64_t)code + 1ull;
    if (code == 1) t[1]++;
    if (code != 2 && fabs(dcrit - 1.) < 1e-9) t[2] |= 1ull;
    if (fabs(conc - 0.1) < 1e-12) t[2] |= 2ull;
    if (code == 2) t[2] |= 4ull;
    t[3]++;
}

void ref_work_destroy(ref_work *w) {
    if (!w) return;
    free(w->delta_x); free(w->surface); free(w->shape_coeff); free(w->B0T);
    free(w->element_mass); free(w->rlmass_matrix); free(w->node_mass); free(w->C_bu);
    free(w->grad_ssh); free(w->grad_terms); free(w->fcor); free(w->VTM); free(w->tmp);
    free(w->D_tau_a); free(w->D_tau_w); free(w->D_del_ci_ridge_myi); free(w->trace);
    free(w);
}

/* model/options.cpp:43,80,111,109,314-376,397,545,547 ; FE.cpp:1167-1172 (turning angle) */
void ref_physical_constants(double out[8]) { /* the constants above, in include/nxs_dyn.h's NXS_CONST_* order */
    out[0] = RHOI; out[1] = RHOW; out[2] = RHOS; out[3] = RHOA; out[4] = GRAVITY; out[5] = OMEGA; out[6] = PI_; out[7] = DAYS_IN_SEC;
}

void ref_default_params(nxs_dyn_params *p) {
    memset(p, 0, sizeof(*p));
    p->dtime_step = 200.;
    p->substeps = 120;
    p->dynamics_type = NXS_DYN_BBM;
    p->basal_stress_type = NXS_BASAL_LEMIEUX;
    p->ice_cat_type = NXS_ICECAT_YOUNG_ICE;
    p->newice_type = 4;
    p->equal_ridging = 0;
    p->use_young_ice_in_myi_reset = 1;
    p->young = 5.9605e+08;
    p->nu0 = 1. / 3.;
    p->tan_phi = 0.7;
    p->compr_strength = 1e10;
    p->compaction_param = -20.;
    p->undamaged_time_relaxation_sigma = 1e7;
    p->exponent_relaxation_sigma = 5.;
    p->compression_factor = 10e3;
    p->exponent_compression_factor = 1.5;
    p->min_h = 0.05;
    p->min_c = 0.01;
    p->quad_drag_coef_water = 0.0055;
    p->lin_drag_coef_water = 0.;
    p->quad_drag_coef_air = 0.0049; /* ASR / constant atmosphere, FE.cpp:1286-1287 */
    p->lin_drag_coef_air = 0.;
    p->ocean_turning_angle_rad = (PI_ / 180.) * 25.;
    p->basal_k1 = 10.;
    p->basal_k2 = 15.;
    p->basal_Cb = 20.;
    p->basal_u_0 = 5e-5;
    p->evp_e = 2.;
    p->evp_Pstar = 27.5e3;
    p->evp_C = 20.;
    p->evp_dmin = 1e-9;
    p->mevp_alpha = 500.;
    p->mevp_beta = 500.;
    p->regrid_angle = 10.;
}

/* FE.cpp:1491-1507 initFETensors */
static void init_fe_tensors(const nxs_dyn_params *p, double Dunit[9]) {
    for (int i = 0; i < 9; ++i) Dunit[i] = 0.;
    double const nu0 = p->nu0;
    double const Dunit_factor = 1. / (1. - nu0 * nu0);
    Dunit[0] = Dunit_factor * 1.;
    Dunit[1] = Dunit_factor * nu0;
    Dunit[3] = Dunit_factor * nu0;
    Dunit[4] = Dunit_factor * 1.;
    Dunit[8] = Dunit_factor * (1. - nu0) / 2.;
}

/* GmshMesh::vertices(indices, um, factor=1.), core/src/gmshmesh.cpp:1929-1939:
 * vertices[i][k] = coords[k] + factor*um[indices[i]-1+k*M_num_nodes] */
static void vertices_um(const nxs_dyn_mesh *m, const double *um, int e, double v[3][2]) {
    int const Nn = m->num_nodes;
    for (int i = 0; i < 3; ++i) {
        int const nd = m->indices[3 * e + i] - 1;
        v[i][0] = m->coord_x[nd];
        v[i][1] = m->coord_y[nd];
        v[i][0] += 1. * um[nd];
        v[i][1] += 1. * um[nd + Nn];
    }
}

/* FE.cpp:1613-1618 jacobian */
static double jacobian(double v[3][2]) {
    double jac = (v[1][0] - v[0][0]) * (v[2][1] - v[0][1]);
    jac -= (v[2][0] - v[0][0]) * (v[1][1] - v[0][1]);
    return jac;
}

/* FE.cpp:1642-1663 sides(element, mesh, um, factor=1) */
static void sides_um(const nxs_dyn_mesh *m, const double *um, int e, double side[3]) {
    double v[3][2];
    vertices_um(m, um, e, v);
    side[0] = hypot(v[1][0] - v[0][0], v[1][1] - v[0][1]);
    side[1] = hypot(v[2][0] - v[1][0], v[2][1] - v[1][1]);
    side[2] = hypot(v[2][0] - v[0][0], v[2][1] - v[0][1]);
}

/* FE.cpp:1929-1933 measure(element, mesh, um) */
static double measure_um(const nxs_dyn_mesh *m, const double *um, int e) {
    double v[3][2];
    vertices_um(m, um, e, v);
    return (1. / 2) * fabs(jacobian(v));
}

/* FE.cpp:1951-1964 shapeCoeff */
static void shape_coeff(const nxs_dyn_mesh *m, const double *um, int e, double coeff[6]) {
    double v[3][2];
    vertices_um(m, um, e, v);
    double const jac = jacobian(v);
    for (int k = 0; k < 3; ++k) {
        int const kp1 = (k + 1) % 3;
        int const kp2 = (k + 2) % 3;
        coeff[k] = (v[kp1][1] - v[kp2][1]) / jac;     /* x derivatives depend on y */
        coeff[k + 3] = (v[kp2][0] - v[kp1][0]) / jac; /* y derivatives depend on x */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* explicitSolve(), part 1: everything that does not change over the sub-time stepping.
 * FE.cpp:10213-10418 ("prep elements" + "prep nodes"). */
void ref_prep(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
              const nxs_dyn_forcing *f, ref_work *w) {
    int const Nn = m->num_nodes, Ne = m->num_elements;
    const double *ssh = f->ssh; /* FE.cpp:10219 */

    init_fe_tensors(p, w->Dunit);

    double const k1 = p->basal_k1, k2 = p->basal_k2, Cb = p->basal_Cb; /* FE.cpp:10208-10211 */

    /* FE.cpp:10230-10234 */
    for (int i = 0; i < Ne; ++i) w->element_mass[i] = 0.;
    for (int i = 0; i < Nn; ++i) { w->rlmass_matrix[i] = 0.; w->node_mass[i] = 0.; w->C_bu[i] = 0.; }
    for (int i = 0; i < 2 * Nn; ++i) w->grad_ssh[i] = 0.;

    for (int cpt = 0; cpt < Ne; ++cpt) { /* FE.cpp:10235 */
        const int32_t *ind = &m->indices[3 * cpt];

        /* Q1, FE.cpp:10239: std::accumulate(..., 0) accumulates into an INT (each partial sum is
         * truncated), then int / size_t -> unsigned integer division by 3. */
        double my_sides[3];
        sides_um(m, s->UM, cpt, my_sides);
        int acc = 0;
        for (int i = 0; i < 3; ++i) acc = (int)(acc + my_sides[i]);
        w->delta_x[cpt] = (double)((unsigned long)acc / (unsigned long)3);

        w->surface[cpt] = measure_um(m, s->UM, cpt); /* FE.cpp:10240 */
        double shapecoeff[6];
        shape_coeff(m, s->UM, cpt, shapecoeff); /* FE.cpp:10241 */
        double *B0T = &w->B0T[18 * cpt];
        for (int i = 0; i < 18; ++i) B0T[i] = 0.;
        for (int i = 0; i < 3; ++i) { /* FE.cpp:10243-10249 */
            B0T[2 * i] = shapecoeff[i];
            B0T[2 * i + 13] = shapecoeff[i];
            B0T[2 * i + 7] = shapecoeff[i + 3];
            B0T[2 * i + 12] = shapecoeff[i + 3];
        }
        for (int i = 0; i < 6; ++i) w->shape_coeff[6 * cpt + i] = shapecoeff[i];

        /* slab mass, FE.cpp:10255-10269 */
        double total_concentration = s->conc[cpt];
        double total_thickness = s->thick[cpt];
        double total_snow = s->snow_thick[cpt];
        if (p->ice_cat_type == NXS_ICECAT_YOUNG_ICE) {
            total_concentration += s->conc_young[cpt];
            total_thickness += s->h_young[cpt];
            total_snow += s->hs_young[cpt];
        }
        if (total_concentration > 0.)
            w->element_mass[cpt] = (RHOI * total_thickness + RHOS * total_snow) / total_concentration;
        else
            w->element_mass[cpt] = 0.;

        /* basal stress, FE.cpp:10273-10308 */
        double element_ssh = 0;
        for (int i = 0; i < 3; ++i) element_ssh += ssh[ind[i] - 1];
        element_ssh /= 3.;

        double max_keel_depth = 28;
        double mean_keel_depth;
        double critical_h = 0.;
        double critical_h_mod = 0.;
        double const min_water_depth = 2.;
        double const depth_eff = STD_MAX(0., element_ssh + STD_MAX(min_water_depth, f->element_depth[cpt]));
        double const g3rd = GRAVITY / 3.;
        switch (p->basal_stress_type) {
        case NXS_BASAL_NONE:
            critical_h = 0.;
            critical_h_mod = 0.;
            break;
        case NXS_BASAL_LEMIEUX:
            mean_keel_depth = k1 * s->thick[cpt];
            mean_keel_depth = STD_MIN(mean_keel_depth, s->conc[cpt] * max_keel_depth);
            critical_h = s->conc[cpt] * depth_eff / k1;
            critical_h_mod = mean_keel_depth / k1;
            break;
        }

        double const element_C_bu = k2 * STD_MAX(0., critical_h_mod - critical_h) * exp(-Cb * (1. - s->conc[cpt]));
        for (int i = 0; i < 3; ++i) { /* FE.cpp:10309-10318 */
            int const idx_node = ind[i] - 1;
            w->rlmass_matrix[idx_node] += w->surface[cpt];
            w->node_mass[idx_node] += w->element_mass[cpt] * w->surface[cpt];
            w->C_bu[idx_node] = STD_MAX(w->C_bu[idx_node], element_C_bu);
        }

        /* gradient of m*g*ssh, FE.cpp:10321-10340.  Q7: the skip test reads node_mass while it is
         * still being accumulated (elements > cpt have not contributed yet). */
        double const m_g_A3rd = w->element_mass[cpt] * w->surface[cpt] * g3rd;
        const double *dxN = &w->shape_coeff[6 * cpt];
        for (int i = 0; i < 3; ++i) {
            int const i_indx = ind[i] - 1;
            if (m->mask_dirichlet[i_indx] || w->node_mass[i_indx] == 0. || m->ghost_nodes[3 * cpt + i])
                continue;
            int const u_indx = i_indx;
            int const v_indx = i_indx + Nn;
            for (int j = 0; j < 3; ++j) {
                int const j_indx = ind[j] - 1;
                w->grad_ssh[u_indx] -= dxN[j] * m_g_A3rd * ssh[j_indx];
                w->grad_ssh[v_indx] -= dxN[j + 3] * m_g_A3rd * ssh[j_indx];
            }
        }
    }

    /* "prep nodes", FE.cpp:10356-10416 */
    int const num_elements = m->nec_width; /* bamgmesh->NodalElementConnectivitySize[1] */
    for (int i = 0; i < Nn; ++i) {
        int const u_indx = i;
        int const v_indx = i + Nn;

        if (w->node_mass[i] == 0.) { /* FE.cpp:10366-10370 */
            s->VT[u_indx] = 0.;
            s->VT[v_indx] = 0.;
        }

        double drag = 0.;
        double surface = 0;
        for (int j = 0; j < num_elements; j++) { /* FE.cpp:10377-10390 */
            /* Q2: the reference casts a NaN pad to int (-> INT_MIN on x86) and skips negatives */
            double const raw = m->nodal_element_connectivity[(size_t)num_elements * i + j];
            if (isnan(raw)) continue;
            int elt_num = (int)(raw - 1);
            if (elt_num < 0) continue;

            double dragp = s->drag_ui[elt_num];
            if (p->ice_cat_type == NXS_ICECAT_YOUNG_ICE && s->conc[elt_num] + s->conc_young[elt_num] > 0.)
                dragp = (s->drag_ui[elt_num] * s->conc[elt_num] + s->drag_ui_young[elt_num] * s->conc_young[elt_num])
                        / (s->conc[elt_num] + s->conc_young[elt_num]);

            drag += dragp * w->surface[elt_num];
            surface += w->surface[elt_num];
        }
        /* Q6: |wind|, not |wind - ice| */
        drag *= RHOA * hypot(f->wind[u_indx], f->wind[v_indx]) / surface;

        w->D_tau_a[u_indx] = drag * f->wind[u_indx];
        w->D_tau_a[v_indx] = drag * f->wind[v_indx];

        w->fcor[i] = 2 * OMEGA * sin(m->lat[i] * PI_ / 180.); /* FE.cpp:10397 */

        w->rlmass_matrix[i] = 1. / w->rlmass_matrix[i]; /* FE.cpp:10400-10402 */
        w->node_mass[i] *= w->rlmass_matrix[i];
        w->rlmass_matrix[i] *= 3.;

        w->VTM[u_indx] = s->VT[u_indx]; /* FE.cpp:10405-10406 */
        w->VTM[v_indx] = s->VT[v_indx];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* FE.cpp:4137-4260 updateSigmaDamage(dt) -- BBM */
void ref_update_sigma_damage(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                             ref_work *w, double const dt) {
    int const Nn = m->num_nodes, Ne = m->num_elements;
    double const nu0 = p->nu0;
    double const sqrt_nu_rhoi = sqrt(2. * (1. + nu0) * RHOI);
    const double min_c = 0.1; /* Q5: hard-coded, NOT dynamics.min_c */
    double *sig0 = s->sigma[0], *sig1 = s->sigma[1], *sig2 = s->sigma[2];
    double *sig[3] = {sig0, sig1, sig2};

    for (int cpt = 0; cpt < Ne; ++cpt) {
        if (s->conc[cpt] <= min_c) { /* FE.cpp:4151-4159 */
            s->damage[cpt] = 0.;
            for (int i = 0; i < 3; i++) sig[i][cpt] = 0.;
            if (w->trace) trace_branch(w->trace + 4 * (size_t)cpt, 2, 0., s->conc[cpt]);
            continue;
        }

        /* FE.cpp:4167-4176 */
        double epsilon_veloc[3] = {0., 0., 0.};
        const double *B0T = &w->B0T[18 * cpt];
        const int32_t *ind = &m->indices[3 * cpt];
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                epsilon_veloc[i] += B0T[i * 6 + 2 * j] * s->VT[ind[j] - 1];
                epsilon_veloc[i] += B0T[i * 6 + 2 * j + 1] * s->VT[ind[j] - 1 + Nn];
            }
        }

        /* FE.cpp:4184-4186 */
        double sigma_n = (sig0[cpt] + sig1[cpt]) * 0.5;
        double const expC = exp(p->compaction_param * (1. - s->conc[cpt]));
        double const time_viscous = p->undamaged_time_relaxation_sigma
                                    * pow((1. - s->damage[cpt]) * expC, p->exponent_relaxation_sigma - 1.);

        /* FE.cpp:4189-4197 */
        double tildeP;
        if (sigma_n < 0.) {
            double const Pmax = pow(s->thick[cpt], p->exponent_compression_factor) * p->compression_factor * expC;
            tildeP = STD_MIN(1., -Pmax / sigma_n);
        } else {
            tildeP = 0.;
        }

        /* Q3, FE.cpp:4199-4200 */
        double const multiplicator = STD_MIN(1. - 1e-12, time_viscous / (time_viscous + dt * (1. - tildeP)));

        double const elasticity = p->young * (1. - s->damage[cpt]) * expC; /* FE.cpp:4202 */

        for (int i = 0; i < 3; i++) { /* FE.cpp:4204-4210 */
            for (int j = 0; j < 3; j++)
                sig[i][cpt] += dt * elasticity * w->Dunit[3 * i + j] * epsilon_veloc[j];
            sig[i][cpt] *= multiplicator;
        }

        /* FE.cpp:4218-4226 */
        double const sigma_s = hypot((sig0[cpt] - sig1[cpt]) / 2., sig2[cpt]);
        sigma_n = (sig0[cpt] + sig1[cpt]) * 0.5;

        double dcrit;
        if (sigma_n < -p->compr_strength)
            dcrit = -p->compr_strength / sigma_n;
        else
            dcrit = s->cohesion[cpt] / (sigma_s + p->tan_phi * sigma_n);

        if (w->trace) trace_branch(w->trace + 4 * (size_t)cpt, ((0. < dcrit) && (dcrit < 1.)) ? 1 : 0, dcrit, s->conc[cpt]);
        if ((0. < dcrit) && (dcrit < 1.)) { /* FE.cpp:4229-4243 */
            double const rtd = sqrt(elasticity) / (w->delta_x[cpt] * sqrt_nu_rhoi);
            double const del_damage = (1.0 - s->damage[cpt]) * (1.0 - dcrit) * dt * rtd;
            s->damage[cpt] += del_damage;
            for (int i = 0; i < 3; i++)
                sig[i][cpt] -= sig[i][cpt] * (1. - dcrit) * dt * rtd;
        }

        /* healing, FE.cpp:4256-4257 */
        s->damage[cpt] = STD_MAX(0., s->damage[cpt]
                                       - dt / s->time_relaxation_damage[cpt] * exp(p->compaction_param * (1. - s->conc[cpt])));
    }
}

/* FE.cpp:10649-10699 updateSigmaVP (EVP: FE.cpp:10705-10715, mEVP: FE.cpp:10721-10726) */
void ref_update_sigma_vp(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                         ref_work *w, double const ralpha1, double const ralpha2) {
    int const Nn = m->num_nodes, Ne = m->num_elements;
    double const e = p->evp_e, Pstar = p->evp_Pstar, C = p->evp_C, delta_min = p->evp_dmin;
    double const re2 = 1. / (e * e);
    double *sig0 = s->sigma[0], *sig1 = s->sigma[1], *sig2 = s->sigma[2];

    for (int cpt = 0; cpt < Ne; cpt++) {
        if (s->thick[cpt] == 0.) {
            sig0[cpt] = 0.; sig1[cpt] = 0.; sig2[cpt] = 0.;
            continue;
        }
        double eps11 = 0., eps22 = 0., eps12 = 0.;
        for (int i = 0; i < 3; i++) {
            double const u = s->VT[m->indices[3 * cpt + i] - 1];
            double const v = s->VT[m->indices[3 * cpt + i] - 1 + Nn];
            double const dxN = w->shape_coeff[6 * cpt + i];
            double const dyN = w->shape_coeff[6 * cpt + i + 3];
            eps11 += dxN * u;
            eps22 += dyN * v;
            eps12 += 0.5 * (dxN * v + dyN * u);
        }
        double const eps1 = eps11 + eps22;
        double const eps2 = eps11 - eps22;

        double const delta = sqrt(eps1 * eps1 + (eps2 * eps2 + 4 * eps12 * eps12) * re2);
        double const P = Pstar * exp(-C * (1. - s->conc[cpt]));
        double const zeta = P / (delta + delta_min);

        double sigma1 = sig0[cpt] + sig1[cpt];
        double sigma2 = sig0[cpt] - sig1[cpt];

        sigma1 += ralpha1 * (zeta * (eps1 - delta) - sigma1);
        sigma2 += ralpha2 * (zeta * eps2 * re2 - sigma2);
        sig2[cpt] += ralpha2 * (zeta * eps12 * re2 - sig2[cpt]);

        sig0[cpt] = 0.5 * (sigma1 + sigma2);
        sig1[cpt] = 0.5 * (sigma1 - sigma2);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* One sub-step up to (not including) updateGhosts: FE.cpp:10425-10530 */
void ref_substep_solve(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                       const nxs_dyn_forcing *f, ref_work *w) {
    int const Nn = m->num_nodes, Ne = m->num_elements;
    int const steps = p->substeps;
    double const dte = p->dtime_step / (double)steps; /* FE.cpp:10185 */
    double const cos_ocean_turning_angle = cos(p->ocean_turning_angle_rad);
    double const sin_ocean_turning_angle = sin(p->ocean_turning_angle_rad);
    double const min_m = RHOI * p->min_h; /* FE.cpp:10191 */
    double const u0 = p->basal_u_0;
    const double *ocean = f->ocean;
    double *VT = s->VT;

    switch (p->dynamics_type) { /* FE.cpp:10427-10440 */
    case NXS_DYN_EVP: {
        double const T = p->dtime_step / 3.;
        double const ralpha1 = 0.5 * dte / T;
        double const ralpha2 = 0.5 * dte / T * p->evp_e * p->evp_e;
        ref_update_sigma_vp(m, p, s, w, ralpha1, ralpha2);
        break;
    }
    case NXS_DYN_MEVP:
        ref_update_sigma_vp(m, p, s, w, 1. / p->mevp_alpha, 1. / p->mevp_alpha);
        break;
    case NXS_DYN_BBM:
        ref_update_sigma_damage(m, p, s, w, dte);
        break;
    default:
        break;
    }

    /* "gradient sigma", FE.cpp:10445-10467 */
    double *grad_terms = w->grad_terms;
    memcpy(grad_terms, w->grad_ssh, sizeof(double) * 2 * (size_t)Nn);
    for (int cpt = 0; cpt < Ne; ++cpt) {
        const double *dxN = &w->shape_coeff[6 * cpt];
        double const volume = s->thick[cpt] * w->surface[cpt];
        for (int i = 0; i < 3; ++i) {
            int const i_indx = m->indices[3 * cpt + i] - 1;
            if (m->mask_dirichlet[i_indx] || w->node_mass[i_indx] == 0. || m->ghost_nodes[3 * cpt + i])
                continue;
            int const u_indx = i_indx;
            int const v_indx = i_indx + Nn;
            grad_terms[u_indx] -= volume * (s->sigma[0][cpt] * dxN[i] + s->sigma[2][cpt] * dxN[i + 3]);
            grad_terms[v_indx] -= volume * (s->sigma[2][cpt] * dxN[i] + s->sigma[1][cpt] * dxN[i + 3]);
        }
    }

    /* "sub-solve", FE.cpp:10472-10529 */
    for (int i = 0; i < m->local_ndof; ++i) {
        if (m->mask_dirichlet[i] || w->node_mass[i] == 0.) continue;

        int u_indx = i;
        int v_indx = i + Nn;

        double dtep, delu, delv;
        if (p->dynamics_type == NXS_DYN_MEVP) {
            double const b_mevp = p->mevp_beta + 1.;
            delu = (w->VTM[u_indx] - VT[u_indx]) / b_mevp;
            delv = (w->VTM[v_indx] - VT[v_indx]) / b_mevp;
            dtep = dte / b_mevp;
        } else {
            delu = 0.;
            delv = 0.;
            dtep = dte;
        }

        double const dte_over_mass = dtep / STD_MAX(min_m, w->node_mass[i]);
        double const uice = VT[u_indx];
        double const vice = VT[v_indx];

        double const c_prime = RHOW * p->quad_drag_coef_water * hypot(ocean[u_indx] - uice, ocean[v_indx] - vice);

        double const tau_b = w->C_bu[i] / (hypot(uice, vice) + u0);
        double const alpha = 1. + dte_over_mass * (c_prime * cos_ocean_turning_angle + tau_b);
        double const beta = dtep * w->fcor[i] + dte_over_mass * c_prime * copysign(sin_ocean_turning_angle, m->lat[i]);
        double const rdenom = 1. / (alpha * alpha + beta * beta);

        double const tau_x = w->D_tau_a[u_indx]
            + c_prime * (ocean[u_indx] * cos_ocean_turning_angle - ocean[v_indx] * copysign(sin_ocean_turning_angle, m->lat[i]));
        double const tau_y = w->D_tau_a[v_indx]
            + c_prime * (ocean[v_indx] * cos_ocean_turning_angle + ocean[u_indx] * copysign(sin_ocean_turning_angle, m->lat[i]));

        double const grad_x = grad_terms[u_indx] * w->rlmass_matrix[i];
        double const grad_y = grad_terms[v_indx] * w->rlmass_matrix[i];

        VT[u_indx] = alpha * uice + beta * vice + dte_over_mass * (alpha * (grad_x + tau_x) + beta * (grad_y + tau_y)) + alpha * delu + beta * delv;
        VT[u_indx] *= rdenom;

        VT[v_indx] = alpha * vice - beta * uice + dte_over_mass * (alpha * (grad_y + tau_y) - beta * (grad_x + tau_x)) + alpha * delv - beta * delu;
        VT[v_indx] *= rdenom;
    }
}

/* "move mesh", FE.cpp:10539-10553 (dt=dte) and FE.cpp:10559-10573 (mEVP, dt=dtime_step) */
void ref_move_mesh(const nxs_dyn_mesh *m, nxs_dyn_state *s, ref_work *w, double const dt) {
    int const n2 = 2 * m->num_nodes;
    double *UM_P = w->tmp;
    memcpy(UM_P, s->UM, sizeof(double) * (size_t)n2);
    for (int nd = 0; nd < n2; ++nd) {
        s->UM[nd] += dt * s->VT[nd];
        s->UT[nd] += dt * s->VT[nd];
    }
    /* M_neumann_nodes = [flag, flag+Nn] pairs, FE.cpp:265-270 */
    for (int k = 0; k < m->num_neumann_flags; ++k) {
        int const nd = m->neumann_flags[k];
        s->UM[nd] = UM_P[nd];
        s->UM[nd + m->num_nodes] = UM_P[nd + m->num_nodes];
    }
}

/* One open-water smoother sweep up to (not including) updateGhosts: FE.cpp:10582-10608.
 * Q8: neighbour order = row order of bamg's NodalConnectivity. */
void ref_smoother_sweep(const nxs_dyn_mesh *m, nxs_dyn_state *s, ref_work *w) {
    int const Nn = m->num_nodes;
    int const max_num_neighbours = m->nc_width;
    double *u = w->tmp;
    double *VT = s->VT;
    memcpy(u, VT, sizeof(double) * 2 * (size_t)Nn);
    for (int i = 0; i < m->local_ndof; ++i) {
        int const u_indx = i;
        int const v_indx = i + Nn;
        if (m->mask_dirichlet[i] || w->node_mass[i] != 0.) continue;

        VT[u_indx] = 0.;
        VT[v_indx] = 0.;

        int num_neighbours = (int)m->nodal_connectivity[(size_t)max_num_neighbours * (i + 1) - 1];
        for (int j = 0; j < num_neighbours; ++j) {
            int const nni = (int)(m->nodal_connectivity[(size_t)max_num_neighbours * i + j] - 1);
            VT[u_indx] += u[nni];
            VT[v_indx] += u[nni + Nn];
        }
        VT[u_indx] /= num_neighbours;
        VT[v_indx] /= num_neighbours;
    }
}

/* FE.cpp:10613-10640: ice-ocean drag diagnostic + mesh move in the open water */
void ref_ow_tail(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                 const nxs_dyn_forcing *f, ref_work *w) {
    int const Nn = m->num_nodes;
    double *UM_P = w->tmp;
    memcpy(UM_P, s->UM, sizeof(double) * 2 * (size_t)Nn);
    for (int i = 0; i < Nn; ++i) {
        int const u_indx = i;
        int const v_indx = i + Nn;

        double const uice = 0.5 * (s->VT[u_indx] + w->VTM[u_indx]);
        double const vice = 0.5 * (s->VT[v_indx] + w->VTM[v_indx]);
        double const c_prime = RHOW * p->quad_drag_coef_water * hypot(f->ocean[u_indx] - uice, f->ocean[v_indx] - vice);
        w->D_tau_w[u_indx] = c_prime * (uice - f->ocean[u_indx]);
        w->D_tau_w[v_indx] = c_prime * (vice - f->ocean[v_indx]);

        if (m->mask_dirichlet[i] || w->node_mass[i] != 0.) continue;

        s->UM[u_indx] += p->dtime_step * s->VT[u_indx];
        s->UM[v_indx] += p->dtime_step * s->VT[v_indx];

        s->UT[u_indx] += p->dtime_step * s->VT[u_indx];
        s->UT[v_indx] += p->dtime_step * s->VT[v_indx];
    }
    for (int k = 0; k < m->num_neumann_flags; ++k) {
        int const nd = m->neumann_flags[k];
        s->UM[nd] = UM_P[nd];
        s->UM[nd + Nn] = UM_P[nd + Nn];
    }
}

/* FE.cpp:10182-10643 explicitSolve() */
void ref_explicit_solve(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                        const nxs_dyn_forcing *f, ref_work *w, ref_ghost_fn ghosts, void *ctx) {
    int const steps = p->substeps;
    double const dte = p->dtime_step / (double)steps;

    ref_prep(m, p, s, f, w);

    for (int st = 0; st < steps; st++) { /* FE.cpp:10423 */
        ref_substep_solve(m, p, s, f, w);
        if (ghosts) ghosts(ctx, s->VT); /* FE.cpp:10534 */
        if (p->dynamics_type != NXS_DYN_MEVP) ref_move_mesh(m, s, w, dte);
    }
    if (p->dynamics_type == NXS_DYN_MEVP) ref_move_mesh(m, s, w, p->dtime_step); /* FE.cpp:10559-10573 */

    /* Q9: 50 sweeps hard-coded, numerics.nit_ow ignored (FE.cpp:10580) */
    for (int nit = 0; nit < 50; ++nit) {
        ref_smoother_sweep(m, s, w);
        if (ghosts) ghosts(ctx, s->VT); /* FE.cpp:10610 */
    }
    ref_ow_tail(m, p, s, f, w);
}

/* ------------------------------------------------------------------------------------------ */
static int bsearch_int(const int32_t *a, int n, int key) { /* std::binary_search */
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = lo + (hi - lo) / 2;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && a[lo] == key);
}

/* FE.cpp:3919-4132 update(UM_P) -- Q9: the UM_P argument is unused by the reference.
 * diffuse(M_sst/M_sss) (FE.cpp:3938-3939) is a no-op at the default diffusivity 0 and those fields
 * are not part of this path. */
void ref_update(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s, ref_work *w) {
    int const Ne = m->num_elements;
    int const equal_ridging = p->equal_ridging;
    int const newice_type = p->newice_type;
    int const use_young_ice_in_myi_reset = p->use_young_ice_in_myi_reset;
    double *D_del = w->D_del_ci_ridge_myi;
    int const young = (p->ice_cat_type == NXS_ICECAT_YOUNG_ICE);

    for (int cpt = 0; cpt < Ne; ++cpt) {
        const int32_t *ind = &m->indices[3 * cpt];
        int to_be_updated = 1; /* FE.cpp:3957-3961 */
        if (bsearch_int(m->neumann_flags, m->num_neumann_flags, ind[0] - 1) ||
            bsearch_int(m->neumann_flags, m->num_neumann_flags, ind[1] - 1) ||
            bsearch_int(m->neumann_flags, m->num_neumann_flags, ind[2] - 1))
            to_be_updated = 0;

        D_del[cpt] = 0.;

        double const surface_old = w->surface[cpt];
        double const old_conc = s->conc[cpt];
        w->surface[cpt] = measure_um(m, s->UM, cpt); /* FE.cpp:3969 */
        if ((s->conc[cpt] > 0.) && to_be_updated) {
            double const surf_ratio = surface_old / w->surface[cpt];
            s->conc[cpt] *= surf_ratio;
            s->thick[cpt] *= surf_ratio;
            s->snow_thick[cpt] *= surf_ratio;
            s->thick_myi[cpt] *= surf_ratio;

            for (int k = 0; k < 3; k++) s->sigma[k][cpt] *= surf_ratio;

            s->ridge_ratio[cpt] = 1. - (1. - s->ridge_ratio[cpt]) * STD_MIN(1., s->conc[cpt]) / (old_conc * surf_ratio);

            if (young) {
                s->h_young[cpt] *= surf_ratio;
                s->conc_young[cpt] *= surf_ratio;
                s->hs_young[cpt] *= surf_ratio;
            }
            if (equal_ridging) {
                double const conc_ratio = STD_MIN(1., s->conc[cpt]) / old_conc;
                s->conc_myi[cpt] *= conc_ratio;
                D_del[cpt] = 0.;
            } else {
                s->conc_myi[cpt] *= surf_ratio;
                D_del[cpt] = -s->conc_myi[cpt];
                s->conc_myi[cpt] = STD_MIN(s->conc_myi[cpt], 1.);
                D_del[cpt] += s->conc_myi[cpt];
            }
            D_del[cpt] *= DAYS_IN_SEC / p->dtime_step;
        }

        /* mechanical redistribution, FE.cpp:4032-4095 */
        double open_water_concentration = 1. - s->conc[cpt];
        if (young) open_water_concentration -= s->conc_young[cpt];
        open_water_concentration = (open_water_concentration < 0.) ? 0. : open_water_concentration;
        open_water_concentration = (open_water_concentration > 1.) ? 1. : open_water_concentration;

        double new_conc_young = 0.;
        double new_h_young = 0.;
        double new_hs_young = 0.;
        double newice = 0.;
        double del_c = 0.;
        double newsnow = 0.;
        double ridge_young_ice_aspect_ratio = 10.;

        if (young) {
            if (s->conc_young[cpt] > 0.) {
                new_conc_young = STD_MIN(1., STD_MAX(0., 1. - s->conc[cpt] - open_water_concentration));

                if ((s->conc[cpt] > p->min_c) && (s->thick[cpt] > p->min_h) && (new_conc_young < s->conc_young[cpt])) {
                    new_h_young = new_conc_young * s->h_young[cpt] / s->conc_young[cpt];
                    new_hs_young = new_conc_young * s->hs_young[cpt] / s->conc_young[cpt];

                    newice = s->h_young[cpt] - new_h_young;
                    del_c = (s->conc_young[cpt] - new_conc_young) / ridge_young_ice_aspect_ratio;
                    newsnow = s->hs_young[cpt] - new_hs_young;

                    s->h_young[cpt] = new_h_young;
                    s->hs_young[cpt] = new_hs_young;

                    s->ridge_ratio[cpt] = 1. - (1. - s->ridge_ratio[cpt]) * s->thick[cpt] / (s->thick[cpt] + newice);
                    s->thick[cpt] += newice;
                    s->snow_thick[cpt] += newsnow;
                }
            } else {
                s->h_young[cpt] = 0.;
                s->hs_young[cpt] = 0.;
            }
        }

        s->conc[cpt] = STD_MIN(1., STD_MAX(0., 1. - new_conc_young - open_water_concentration + del_c));
        if (young) {
            new_conc_young = STD_MAX(0., STD_MIN(new_conc_young, 1. - s->conc[cpt]));
            s->conc_young[cpt] = new_conc_young;
        }

        double max_true_thickness = 50.; /* FE.cpp:4098-4110 */
        if (s->conc[cpt] > 0.) {
            double test_h_thick = s->thick[cpt] / s->conc[cpt];
            test_h_thick = (test_h_thick > max_true_thickness) ? max_true_thickness : test_h_thick;
            s->conc[cpt] = STD_MIN(1. - new_conc_young, s->thick[cpt] / test_h_thick);
        } else {
            s->ridge_ratio[cpt] = 0.;
            s->thick[cpt] = 0.;
            s->snow_thick[cpt] = 0.;
        }

        /* lower bounds, FE.cpp:4120-4130 */
        s->conc[cpt] = ((s->conc[cpt] > 0.) ? (s->conc[cpt]) : (0.));
        s->thick[cpt] = ((s->thick[cpt] > 0.) ? (s->thick[cpt]) : (0.));
        s->thick_myi[cpt] = ((s->thick_myi[cpt] > 0.) ? (s->thick_myi[cpt]) : (0.));
        s->snow_thick[cpt] = ((s->snow_thick[cpt] > 0.) ? (s->snow_thick[cpt]) : (0.));
        D_del[cpt] = -s->conc_myi[cpt];
        if (newice_type == 4 && use_young_ice_in_myi_reset)
            s->conc_myi[cpt] = STD_MAX(0., STD_MIN(s->conc_myi[cpt], s->conc[cpt] + s->conc_young[cpt]));
        else
            s->conc_myi[cpt] = STD_MAX(0., STD_MIN(s->conc_myi[cpt], s->conc[cpt]));
        D_del[cpt] += s->conc_myi[cpt];
    }
}

/* FE.cpp:10140-10176 updateFreeDriftVelocity */
void ref_free_drift(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
                    const nxs_dyn_forcing *f) {
    int const Nn = m->num_nodes;
    double norm_Voce_ice, coef_Voce, norm_Vair_ice, coef_Vair;
    double norm_Voce_ice_min = 0.01;
    double norm_Vair_ice_min = 0.01;
    double *VT = s->VT;
    for (int nd = 0; nd < Nn; ++nd) {
        if (!m->mask_dirichlet[nd]) {
            int index_u = nd;
            int index_v = nd + Nn;

            norm_Voce_ice = hypot(VT[index_u] - f->ocean[index_u], VT[index_v] - f->ocean[index_v]);
            norm_Voce_ice = (norm_Voce_ice > norm_Voce_ice_min) ? (norm_Voce_ice) : norm_Voce_ice_min;

            coef_Voce = p->lin_drag_coef_water + p->quad_drag_coef_water * norm_Voce_ice;
            coef_Voce *= RHOW;

            norm_Vair_ice = hypot(VT[index_u] - f->wind[index_u], VT[index_v] - f->wind[index_v]);
            norm_Vair_ice = (norm_Vair_ice > norm_Vair_ice_min) ? (norm_Vair_ice) : norm_Vair_ice_min;

            coef_Vair = p->lin_drag_coef_air + p->quad_drag_coef_air * norm_Vair_ice;
            coef_Vair *= (RHOA);

            VT[index_u] = (coef_Vair * f->wind[index_u] + coef_Voce * f->ocean[index_u]) / (coef_Vair + coef_Voce);
            VT[index_v] = (coef_Vair * f->wind[index_v] + coef_Voce * f->ocean[index_v]) / (coef_Vair + coef_Voce);

            s->UT[index_u] += p->dtime_step * VT[index_u];
            s->UT[index_v] += p->dtime_step * VT[index_v];
        }
    }
}

/* FE.cpp:8197-8214 */
void ref_step(const nxs_dyn_mesh *m, const nxs_dyn_params *p, nxs_dyn_state *s,
              const nxs_dyn_forcing *f, ref_work *w, ref_ghost_fn ghosts, void *ctx) {
    if (p->dynamics_type == NXS_DYN_FREE_DRIFT) {
        ref_free_drift(m, p, s, f);
    } else if (p->dynamics_type != NXS_DYN_NO_MOTION) {
        ref_explicit_solve(m, p, s, f, w, ghosts, ctx);
        ref_update(m, p, s, w);
    }
}

/* ------------------------------------------------------------------------------------------ */
static void sort3(double a[3]) { /* std::sort on 3 values */
    double t;
    if (a[1] < a[0]) { t = a[0]; a[0] = a[1]; a[1] = t; }
    if (a[2] < a[1]) { t = a[1]; a[1] = a[2]; a[2] = t; }
    if (a[1] < a[0]) { t = a[0]; a[0] = a[1]; a[1] = t; }
}

/* FE.cpp:8298-8309 checkRegridding -> minAngle (FE.cpp:1795-1816), minAngles (FE.cpp:1758-1768),
 * flip (FE.cpp:1824-1839).  Local part only; the all_reduce is the caller's. */
int ref_check_regridding(const nxs_dyn_mesh *m, const nxs_dyn_params *p, const nxs_dyn_state *s,
                         double *min_angle_out, int32_t *flip_out) {
    int const Ne = m->num_elements;
    double min_angle = INFINITY, minarea = INFINITY, maxarea = -INFINITY;
    for (int cpt = 0; cpt < Ne; ++cpt) {
        double side[3];
        sides_um(m, s->UM, cpt, side);
        sort3(side);
        double minang = acos((pow(side[1], 2.) + pow(side[2], 2.) - pow(side[0], 2.)) / (2 * side[1] * side[2]));
        minang = minang * 45.0 / atan(1.0);
        if (cpt == 0 || minang < min_angle) min_angle = minang; /* std::min_element */

        double v[3][2];
        vertices_um(m, s->UM, cpt, v);
        double const jac = jacobian(v);
        if (cpt == 0 || jac < minarea) minarea = jac;
        if (cpt == 0 || maxarea < jac) maxarea = jac;
    }
    int const flip = ((minarea <= 0.) && (maxarea >= 0.));
    if (min_angle_out) *min_angle_out = min_angle;
    if (flip_out) *flip_out = flip;
    return (min_angle < p->regrid_angle) || flip;
}

/* FE.cpp:14536-14655 checkFieldsFast -- restricted to the fields of this path (M_tice, M_sst,
 * M_sss, M_tsurf_young belong to thermo and are not part of nxs_dyn_state). Returns crash flag. */
static int out_of_range(const double *a, int n, double lo, double hi) {
    for (int i = 0; i < n; i++) {
        double val = a[i];
        if (val > hi) return 1;
        if (val < lo) return 1;
        if (isnan(val)) return 1;
    }
    return 0;
}

int ref_check_fields_fast(const nxs_dyn_mesh *m, const nxs_dyn_params *p, const nxs_dyn_state *s) {
    int const Ne = m->num_elements, Nn = m->num_nodes;
    int crash = 0;
    crash |= out_of_range(s->thick, Ne, 0., 50.);
    crash |= out_of_range(s->snow_thick, Ne, 0., 10.);
    crash |= out_of_range(s->conc, Ne, 0., 1.);
    crash |= out_of_range(s->damage, Ne, 0., 1.);
    crash |= out_of_range(s->ridge_ratio, Ne, 0., 1.);
    if (p->ice_cat_type == NXS_ICECAT_YOUNG_ICE) {
        crash |= out_of_range(s->h_young, Ne, 0., 2.);
        crash |= out_of_range(s->hs_young, Ne, 0., 2.);
        crash |= out_of_range(s->conc_young, Ne, 0., 1.);
    }
    for (int i = 0; i < Nn; i++) {
        if (hypot(s->VT[i], s->VT[i + Nn]) > 5.) { crash = 1; break; }
        if (isnan(s->VT[i] + s->VT[i + Nn])) { crash = 1; break; }
    }
    return crash;
}

/* ------------------------------------------------------------------------------------------ */
/* FE.cpp:13967-13977 (pack) and :13987-13995 (unpack): [u-block | v-block] per neighbour */
void ref_ghosts_pack(const nxs_dyn_halo *h, int32_t Nn, const double *vec, int k, double *buf) {
    int const off = h->send_offsets[k];
    int const srl = h->send_offsets[k + 1] - off;
    for (int j = 0; j < srl; j++) {
        buf[j] = vec[h->send_index[off + j]];
        buf[j + srl] = vec[h->send_index[off + j] + Nn];
    }
}

void ref_ghosts_unpack(const nxs_dyn_halo *h, int32_t Nn, double *vec, int k, const double *buf) {
    int const off = h->recv_offsets[k];
    int const srl = h->recv_offsets[k + 1] - off;
    for (int j = 0; j < srl; j++) {
        vec[h->recv_index[off + j]] = buf[j];
        vec[h->recv_index[off + j] + Nn] = buf[j + srl];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Restatement of the two connectivity tables Mesh::WriteMesh builds
 * (contrib/bamg/src/Mesh.cpp:514-543 chains, :579-630 IssmEdges, :798-865 tables) for a mesh that
 * came through BamgConvertMeshx (all triangles are "inside" triangles in input order).
 *
 *  - NodalElementConnectivity row v: elements holding v, newest chain entry first, i.e. DESCENDING
 *    element number (Mesh.cpp:526-538, 804-811); NaN padded; 1-based.
 *  - NodalConnectivity row v: the other end of every edge holding v, newest chain entry first,
 *    where edges are numbered by first appearance when walking triangles in order and their edges
 *    in the order VerticesOfTriangularEdge = {1,2},{2,0},{0,1} (macros.h:13; Mesh.cpp:586-605,
 *    830-839, 850-865); 0 padded; last column = count; 1-based.
 */
int ref_mesh_connectivity(const int32_t *indices, int32_t Nn, int32_t Ne,
                          int32_t *nec_width, double *nec, int32_t *nc_width, double *nc) {
    static const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
    if (Nn <= 0 || Ne <= 0) return -1;

    /* chains for the element fan */
    int *head_1 = (int *)malloc(sizeof(int) * (size_t)Nn);
    int *next_1 = (int *)malloc(sizeof(int) * 3 * (size_t)Ne);
    int *size_1 = (int *)calloc((size_t)Nn, sizeof(int));
    for (int i = 0; i < Nn; i++) head_1[i] = -1;
    int k = 0;
    for (int i = 0; i < Ne; i++)
        for (int j = 0; j < 3; j++) {
            int v = indices[3 * i + j] - 1;
            next_1[k] = head_1[v];
            head_1[v] = k++;
            size_1[v] += 1;
        }
    int max_1 = 0;
    for (int i = 0; i < Nn; i++) if (size_1[i] > max_1) max_1 = size_1[i];
    if (nec_width) *nec_width = max_1;
    if (nec) {
        for (size_t i = 0; i < (size_t)max_1 * Nn; i++) nec[i] = NAN;
        for (int i = 0; i < Nn; i++) {
            int kk = 0;
            for (int j = head_1[i]; j != -1; j = next_1[j]) {
                nec[(size_t)max_1 * i + kk] = floor((double)j / 3) + 1;
                kk++;
            }
        }
    }

    /* unique edges in order of first appearance (SetOfEdges4 hashed on min vertex) */
    long nbax = 3L * Ne;
    int *e_i = (int *)malloc(sizeof(int) * (size_t)nbax);
    int *e_j = (int *)malloc(sizeof(int) * (size_t)nbax);
    int *e_next = (int *)malloc(sizeof(int) * (size_t)nbax);
    int *e_first = (int *)malloc(sizeof(int) * (size_t)nbax); /* first element holding the edge */
    int *e_head = (int *)malloc(sizeof(int) * (size_t)Nn);
    for (int i = 0; i < Nn; i++) e_head[i] = -1;
    int nbe = 0;
    for (int t = 0; t < Ne; t++)
        for (int j = 0; j < 3; j++) {
            int i1 = indices[3 * t + VOTE[j][0]] - 1;
            int i2 = indices[3 * t + VOTE[j][1]] - 1;
            int a = i1 <= i2 ? i1 : i2, b = i1 <= i2 ? i2 : i1;
            int n = e_head[a];
            while (n >= 0) {
                if (e_i[n] == a && e_j[n] == b) break;
                n = e_next[n];
            }
            if (n < 0) {
                e_i[nbe] = a; e_j[nbe] = b; e_next[nbe] = e_head[a]; e_head[a] = nbe;
                e_first[nbe] = t;
                nbe++;
            }
        }
    /* IssmEdges[i][0..1]: oriented as in the first element holding the edge (Mesh.cpp:609-627) */
    int *ie0 = (int *)malloc(sizeof(int) * (size_t)nbe);
    int *ie1 = (int *)malloc(sizeof(int) * (size_t)nbe);
    for (int i = 0; i < nbe; i++) {
        int t = e_first[i];
        ie0[i] = e_i[i] + 1; ie1[i] = e_j[i] + 1;
        for (int j = 0; j < 3; j++) {
            if (indices[3 * t + j] - 1 == e_i[i]) {
                if (indices[3 * t + (j + 1) % 3] - 1 == e_j[i]) { ie0[i] = e_i[i] + 1; ie1[i] = e_j[i] + 1; }
                else { ie0[i] = e_j[i] + 1; ie1[i] = e_i[i] + 1; }
                break;
            }
        }
    }
    /* chains over edge ends */
    int *head_2 = (int *)malloc(sizeof(int) * (size_t)Nn);
    int *next_2 = (int *)malloc(sizeof(int) * 2 * (size_t)nbe);
    int *size_2 = (int *)calloc((size_t)Nn, sizeof(int));
    for (int i = 0; i < Nn; i++) head_2[i] = -1;
    k = 0;
    for (int i = 0; i < nbe; i++)
        for (int j = 0; j < 2; j++) {
            int v = (j == 0 ? ie0[i] : ie1[i]) - 1;
            next_2[k] = head_2[v];
            head_2[v] = k++;
            size_2[v] += 1;
        }
    int max_2 = 0;
    for (int i = 0; i < Nn; i++) if (size_2[i] > max_2) max_2 = size_2[i];
    max_2++; /* last column holds the count */
    if (nc_width) *nc_width = max_2;
    if (nc) {
        for (size_t i = 0; i < (size_t)max_2 * Nn; i++) nc[i] = 0;
        for (int i = 0; i < Nn; i++) {
            int kk = 0;
            for (int j = head_2[i]; j != -1; j = next_2[j]) {
                int num = ie0[j / 2];
                if (i + 1 == num) nc[(size_t)max_2 * i + kk] = ie1[j / 2];
                else nc[(size_t)max_2 * i + kk] = num;
                kk++;
            }
            nc[(size_t)max_2 * (i + 1) - 1] = kk;
        }
    }
    free(head_1); free(next_1); free(size_1);
    free(e_i); free(e_j); free(e_next); free(e_first); free(e_head);
    free(ie0); free(ie1); free(head_2); free(next_2); free(size_2);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* updateIceDiagnostics(), FE.cpp:7860-7905 (D_tsurf and the FSD parameters excepted: thermodynamic / OASIS variables).  out[k] may be NULL. */
void ref_ice_diagnostics(const nxs_dyn_mesh *m, const nxs_dyn_params *p, const nxs_dyn_state *s, double *D_conc, double *D_thick,
                         double *D_snow_thick, double *D_sigma0, double *D_sigma1, double *D_divergence) {
    int const Nn = m->num_nodes;
    for (int i = 0; i < m->num_elements; i++) {
        double dc = s->conc[i], dt = s->thick[i], ds = s->snow_thick[i];   /* :7872-7874 */
        if (p->ice_cat_type == NXS_ICECAT_YOUNG_ICE) {                       /* :7876-7882 */
            dc += s->conc_young[i];
            dt += s->h_young[i];
            ds += s->hs_young[i];
        }
        if (D_conc) D_conc[i] = dc;
        if (D_thick) D_thick[i] = dt;
        if (D_snow_thick) D_snow_thick[i] = ds;
        /* principal stresses, :7885-7887 */
        if (D_sigma0) D_sigma0[i] = (s->sigma[0][i] + s->sigma[1][i]) / 2.;
        if (D_sigma1) D_sigma1[i] = hypot((s->sigma[0][i] - s->sigma[1][i]) / 2., s->sigma[2][i]);
        /* divergence, :7889-7900: shapeCoeff on the mesh displaced by M_UM (FE.cpp:1951-1964, 1613-1618) */
        if (D_divergence) {
            double vx[3], vy[3];
            for (int j = 0; j < 3; j++) {
                int const n = m->indices[3 * i + j] - 1;
                vx[j] = m->coord_x[n] + 1. * s->UM[n];
                vy[j] = m->coord_y[n] + 1. * s->UM[n + Nn];
            }
            double jac = (vx[1] - vx[0]) * (vy[2] - vy[0]);
            jac -= (vx[2] - vx[0]) * (vy[1] - vy[0]);
            double div = 0.;
            for (int j = 0; j < 3; j++) {
                int const n = m->indices[3 * i + j] - 1;
                int const kp1 = (j + 1) % 3, kp2 = (j + 2) % 3;
                double const dxN = (vy[kp1] - vy[kp2]) / jac, dyN = (vx[kp2] - vx[kp1]) / jac;
                double const u = s->VT[n], v = s->VT[n + Nn];
                div += dxN * u + dyN * v;
            }
            D_divergence[i] = div;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* P in-process ranks stepped in lock-step by a pool of threads: the CPU analogue of the reference's MPI run (one partition per
 * core, every updateGhosts(M_VT) of FE.cpp:10425-10611 a shared-memory exchange between two barriers), used by bench.py's
 * cpu_baseline leg and checked against the serial multirank_step of oracle/pyoracle.py (tests/test_multirank_oracle.py).
 * Thread t runs the ranks t, t + nthreads, ...; the phases are those of ref_explicit_solve / ref_step. */
#include <pthread.h>

typedef struct ref_mr_job {
    int nranks, nthreads, nsteps;
    const nxs_dyn_mesh *const *m;
    const nxs_dyn_params *p;
    nxs_dyn_state *const *s;
    const nxs_dyn_forcing *const *f;
    ref_work *const *w;
    const nxs_dyn_halo *const *h;
    double **sendbuf;       /* [nranks] rank r's packed segments, segment k at 2 * send_offsets[k] */
    int **peer_seg;         /* [nranks][num_recv_procs] the index k' of me among the sender's send_procs */
    pthread_barrier_t bar;
} ref_mr_job;

typedef struct ref_mr_arg { ref_mr_job *job; int tid; } ref_mr_arg;

static void mr_pack(ref_mr_job *j, int r) {
    const nxs_dyn_halo *h = j->h[r];
    for (int k = 0; k < h->num_send_procs; k++)
        ref_ghosts_pack(h, j->m[r]->num_nodes, j->s[r]->VT, k, j->sendbuf[r] + 2 * (size_t)h->send_offsets[k]);
}
static void mr_unpack(ref_mr_job *j, int r) {
    const nxs_dyn_halo *h = j->h[r];
    for (int k = 0; k < h->num_recv_procs; k++) {
        const int q = h->recv_procs[k], kk = j->peer_seg[r][k];
        ref_ghosts_unpack(h, j->m[r]->num_nodes, j->s[r]->VT, k, j->sendbuf[q] + 2 * (size_t)j->h[q]->send_offsets[kk]);
    }
}

#define MR_EACH(stmt) do { for (int r = tid; r < J->nranks; r += J->nthreads) { stmt; } pthread_barrier_wait(&J->bar); } while (0)

static void *mr_thread(void *arg_) {
    ref_mr_arg *a = (ref_mr_arg *)arg_;
    ref_mr_job *J = a->job;
    const int tid = a->tid;
    const nxs_dyn_params *p = J->p;
    const int S = p->substeps;
    const double dte = p->dtime_step / (double)S;
    for (int it = 0; it < J->nsteps; it++) {
        if (p->dynamics_type == NXS_DYN_NO_MOTION) break;
        if (p->dynamics_type == NXS_DYN_FREE_DRIFT) { MR_EACH(ref_free_drift(J->m[r], p, J->s[r], J->f[r])); continue; }
        MR_EACH(ref_prep(J->m[r], p, J->s[r], J->f[r], J->w[r]));
        for (int ss = 0; ss < S; ss++) {
            MR_EACH(ref_substep_solve(J->m[r], p, J->s[r], J->f[r], J->w[r]); mr_pack(J, r));
            if (p->dynamics_type != NXS_DYN_MEVP) MR_EACH(mr_unpack(J, r); ref_move_mesh(J->m[r], J->s[r], J->w[r], dte));
            else MR_EACH(mr_unpack(J, r));
        }
        if (p->dynamics_type == NXS_DYN_MEVP) MR_EACH(ref_move_mesh(J->m[r], J->s[r], J->w[r], p->dtime_step));
        for (int nit = 0; nit < 50; nit++) {  /* Q9 */
            MR_EACH(ref_smoother_sweep(J->m[r], J->s[r], J->w[r]); mr_pack(J, r));
            MR_EACH(mr_unpack(J, r));
        }
        MR_EACH(ref_ow_tail(J->m[r], p, J->s[r], J->f[r], J->w[r]); ref_update(J->m[r], p, J->s[r], J->w[r]));
    }
    return NULL;
}

/* 0 on success; -1: bad arguments / inconsistent halo lists; -2: out of memory / threads */
int ref_multirank_steps(int nranks, const nxs_dyn_mesh *const *m, const nxs_dyn_params *p, nxs_dyn_state *const *s,
                        const nxs_dyn_forcing *const *f, ref_work *const *w, const nxs_dyn_halo *const *h, int nsteps, int nthreads) {
    if (nranks < 1 || nsteps < 0 || !m || !p || !s || !f || !w || !h) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nranks) nthreads = nranks;
    ref_mr_job J;
    memset(&J, 0, sizeof J);
    J.nranks = nranks; J.nthreads = nthreads; J.nsteps = nsteps;
    J.m = m; J.p = p; J.s = s; J.f = f; J.w = w; J.h = h;
    int rc = 0;
    J.sendbuf = (double **)calloc((size_t)nranks, sizeof(double *));
    J.peer_seg = (int **)calloc((size_t)nranks, sizeof(int *));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    ref_mr_arg *args = (ref_mr_arg *)calloc((size_t)nthreads, sizeof(ref_mr_arg));
    if (!J.sendbuf || !J.peer_seg || !th || !args) rc = -2;
    for (int r = 0; r < nranks && rc == 0; r++) {
        const nxs_dyn_halo *hr = h[r];
        const int ns = hr->num_send_procs, nr = hr->num_recv_procs;
        J.sendbuf[r] = (double *)malloc(sizeof(double) * (2 * (size_t)(ns ? hr->send_offsets[ns] : 0) + 1));
        J.peer_seg[r] = (int *)malloc(sizeof(int) * (size_t)(nr + 1));
        if (!J.sendbuf[r] || !J.peer_seg[r]) { rc = -2; break; }
        for (int k = 0; k < nr && rc == 0; k++) {
            const int q = hr->recv_procs[k];
            if (q < 0 || q >= nranks) { rc = -1; break; }
            int found = -1;
            for (int kk = 0; kk < h[q]->num_send_procs; kk++) if (h[q]->send_procs[kk] == hr->rank) found = kk;
            if (found < 0 || h[q]->send_offsets[found + 1] - h[q]->send_offsets[found] != hr->recv_offsets[k + 1] - hr->recv_offsets[k]) { rc = -1; break; }
            J.peer_seg[r][k] = found;
        }
    }
    if (rc == 0 && pthread_barrier_init(&J.bar, NULL, (unsigned)nthreads) != 0) rc = -2;
    if (rc == 0) {
        int started = 0;
        for (int t = 0; t < nthreads; t++) {
            args[t].job = &J; args[t].tid = t;
            if (t == nthreads - 1) { mr_thread(&args[t]); }          /* the caller's thread works too */
            else if (pthread_create(&th[t], NULL, mr_thread, &args[t]) != 0) { rc = -2; break; }  /* (a missing thread would hang the barrier: cannot happen past this point) */
            else started++;
        }
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
        pthread_barrier_destroy(&J.bar);
    }
    if (J.sendbuf) for (int r = 0; r < nranks; r++) free(J.sendbuf[r]);
    if (J.peer_seg) for (int r = 0; r < nranks; r++) free(J.peer_seg[r]);
    free(J.sendbuf); free(J.peer_seg); free(th); free(args);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* The same lock-step run as a PERSISTENT context that uses the host as an MPI run of the reference would (bench.py's cpu_baseline; SURVEY 8d
 * "(ii) all physical cores"):
 *   - the threads live as long as the context; thread t is pinned to one CPU (pthread_setaffinity_np), physical cores before their SMT siblings and the
 *     sockets taken in turn, so that n threads use the memory controllers of every socket;
 *   - every partition's arrays (mesh, state, forcing, halo lists, work) are ALLOCATED AND FIRST TOUCHED BY THE THREAD THAT OWNS THE PARTITION: its pages
 *     lie on that thread's NUMA node, like the heap of an MPI rank (in round 3 the Python main thread had touched them all: one node served 256 threads);
 *   - the phases meet at a sense-reversing spin barrier instead of a futex one (340 meetings per step).
 * ref_mr_run's results are bit for bit those of ref_multirank_steps (tests/test_multirank_oracle.py); ref_mr_destroy copies the state back. */
#include <sched.h>
#include <stdatomic.h>
#include <stdio.h>

typedef struct ref_mr_rank {
    nxs_dyn_mesh m; nxs_dyn_state s; nxs_dyn_forcing f; nxs_dyn_halo h; ref_work *w;
    void **owned; int n_owned, cap_owned;   /* every array this context allocated for the rank */
} ref_mr_rank;

struct ref_mr_ctx {
    int nranks, nthreads, pin;
    nxs_dyn_params p;
    const nxs_dyn_mesh *const *src_m; nxs_dyn_state *const *src_s; const nxs_dyn_forcing *const *src_f; const nxs_dyn_halo *const *src_h;
    ref_mr_rank *rk;
    ref_mr_job job;                /* the arrays of pointers mr_thread's phases read (m, s, f, w, h, sendbuf, peer_seg) */
    const nxs_dyn_mesh **pm; nxs_dyn_state **ps; const nxs_dyn_forcing **pf; ref_work **pw; const nxs_dyn_halo **ph;
    pthread_t *th;
    int *cpu_of;                   /* [nthreads] the CPU thread t is pinned to, -1: not pinned */
    int *cpu_plan;                 /* [nthreads] the CPU thread t WOULD be pinned to (ref_mr_configure switches pinning on and off between runs) */
    cpu_set_t all_cpus;            /* the affinity mask the process came with (what an unpinned thread runs on) */
    int barrier_kind;              /* 0: sense-reversing spin barrier; 1: pthread_barrier_t (futex: the waiting threads sleep -- under a CPU quota spinning burns the
                                    * quota the working threads need, round 4: 16 pinned spinning threads 1.8e8 against 2.85e8 for round 3's sleeping ones) */
    pthread_barrier_t pbar;
    int sockets_used, physical_cores_used;
    /* command hand-off (idle threads sleep on the condition variable; inside a run they meet at the spin barrier) */
    pthread_mutex_t mu; pthread_cond_t cv_go, cv_done;
    int command, generation, done, failed;   /* command: 1 = set up (copy in), 2 = run, 3 = copy back, 4 = exit */
    int nsteps;
    atomic_int bar_count; atomic_int bar_sense;
};
typedef struct ref_mr_ctx ref_mr_ctx;
typedef struct ref_mr_targ { ref_mr_ctx *c; int tid; } ref_mr_targ;

static void mr_spin_barrier(ref_mr_ctx *c, int *local_sense) {
    if (c->barrier_kind == 1) { pthread_barrier_wait(&c->pbar); return; }
    const int sense = !*local_sense;
    *local_sense = sense;
    if (atomic_fetch_add_explicit(&c->bar_count, 1, memory_order_acq_rel) == c->nthreads - 1) {
        atomic_store_explicit(&c->bar_count, 0, memory_order_relaxed);
        atomic_store_explicit(&c->bar_sense, sense, memory_order_release);
    } else {
        int spins = 0;
        while (atomic_load_explicit(&c->bar_sense, memory_order_acquire) != sense) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
            if (++spins > 20000) { sched_yield(); spins = 0; }   /* (more threads than CPUs: let the others run) */
        }
    }
}

static void *mr_own(ref_mr_rank *r, const void *src, size_t bytes) {   /* a copy made -- and so first touched -- by the calling thread */
    if (!src) return NULL;
    void *p = malloc(bytes ? bytes : 1);
    if (!p) return NULL;
    if (bytes) memcpy(p, src, bytes);
    if (r->n_owned == r->cap_owned) {
        const int cap = r->cap_owned ? 2 * r->cap_owned : 64;
        void **q = (void **)realloc(r->owned, (size_t)cap * sizeof(void *));
        if (!q) { free(p); return NULL; }
        r->owned = q; r->cap_owned = cap;
    }
    r->owned[r->n_owned++] = p;
    return p;
}

static int mr_copy_in(ref_mr_ctx *c, int r) {
    ref_mr_rank *k = &c->rk[r];
    const nxs_dyn_mesh *m = c->src_m[r];
    const nxs_dyn_state *s = c->src_s[r];
    const nxs_dyn_forcing *f = c->src_f[r];
    const nxs_dyn_halo *h = c->src_h[r];
    const size_t Nn = (size_t)m->num_nodes, Ne = (size_t)m->num_elements, D = sizeof(double);
    int bad = 0;
#define OWN(dst, src, bytes) do { (dst) = mr_own(k, (src), (bytes)); if ((src) && !(dst)) bad = 1; } while (0)
    k->m = *m;
    OWN(k->m.indices, m->indices, 3 * Ne * sizeof(int32_t)); OWN(k->m.ghost_nodes, m->ghost_nodes, 3 * Ne);
    OWN(k->m.coord_x, m->coord_x, Nn * D); OWN(k->m.coord_y, m->coord_y, Nn * D); OWN(k->m.lat, m->lat, Nn * D);
    OWN(k->m.mask_dirichlet, m->mask_dirichlet, Nn); OWN(k->m.neumann_flags, m->neumann_flags, (size_t)m->num_neumann_flags * sizeof(int32_t));
    OWN(k->m.nodal_element_connectivity, m->nodal_element_connectivity, Nn * (size_t)m->nec_width * D);
    OWN(k->m.nodal_connectivity, m->nodal_connectivity, Nn * (size_t)m->nc_width * D);
    k->s = *s;
    OWN(k->s.VT, s->VT, 2 * Nn * D); OWN(k->s.UM, s->UM, 2 * Nn * D); OWN(k->s.UT, s->UT, 2 * Nn * D);
    OWN(k->s.conc, s->conc, Ne * D); OWN(k->s.thick, s->thick, Ne * D); OWN(k->s.snow_thick, s->snow_thick, Ne * D);
    OWN(k->s.damage, s->damage, Ne * D); OWN(k->s.ridge_ratio, s->ridge_ratio, Ne * D);
    for (int i = 0; i < 3; i++) OWN(k->s.sigma[i], s->sigma[i], Ne * D);
    OWN(k->s.conc_young, s->conc_young, Ne * D); OWN(k->s.h_young, s->h_young, Ne * D); OWN(k->s.hs_young, s->hs_young, Ne * D);
    OWN(k->s.conc_myi, s->conc_myi, Ne * D); OWN(k->s.thick_myi, s->thick_myi, Ne * D);
    OWN(k->s.cohesion, s->cohesion, Ne * D); OWN(k->s.time_relaxation_damage, s->time_relaxation_damage, Ne * D);
    OWN(k->s.drag_ui, s->drag_ui, Ne * D); OWN(k->s.drag_ui_young, s->drag_ui_young, Ne * D);
    k->f = *f;
    OWN(k->f.wind, f->wind, 2 * Nn * D); OWN(k->f.ocean, f->ocean, 2 * Nn * D); OWN(k->f.ssh, f->ssh, Nn * D); OWN(k->f.element_depth, f->element_depth, Ne * D);
    k->h = *h;
    const size_t ns = (size_t)h->num_send_procs, nr = (size_t)h->num_recv_procs;
    OWN(k->h.send_procs, h->send_procs, ns * sizeof(int32_t)); OWN(k->h.send_offsets, h->send_offsets, (ns + 1) * sizeof(int32_t));
    OWN(k->h.send_index, h->send_index, (size_t)(ns ? h->send_offsets[ns] : 0) * sizeof(int32_t));
    OWN(k->h.recv_procs, h->recv_procs, nr * sizeof(int32_t)); OWN(k->h.recv_offsets, h->recv_offsets, (nr + 1) * sizeof(int32_t));
    OWN(k->h.recv_index, h->recv_index, (size_t)(nr ? h->recv_offsets[nr] : 0) * sizeof(int32_t));
#undef OWN
    k->w = ref_work_create(m->num_nodes, m->num_elements);   /* calloc: touched by this thread at its first step */
    if (!k->w) bad = 1;
    {   /* the exchange buffer of the rank (what ref_multirank_steps allocates on the caller's thread) */
        const size_t n = 2 * (size_t)(ns ? h->send_offsets[ns] : 0) + 1;
        c->job.sendbuf[r] = (double *)malloc(n * D);
        if (c->job.sendbuf[r]) memset(c->job.sendbuf[r], 0, n * D); else bad = 1;
    }
    c->pm[r] = &k->m; c->ps[r] = &k->s; c->pf[r] = &k->f; c->pw[r] = k->w; c->ph[r] = &k->h;
    return bad;
}

static void mr_copy_back(ref_mr_ctx *c, int r) {
    ref_mr_rank *k = &c->rk[r];
    nxs_dyn_state *s = c->src_s[r];
    const size_t Nn = (size_t)k->m.num_nodes, Ne = (size_t)k->m.num_elements, D = sizeof(double);
    memcpy(s->VT, k->s.VT, 2 * Nn * D); memcpy(s->UM, k->s.UM, 2 * Nn * D); memcpy(s->UT, k->s.UT, 2 * Nn * D);
    double *dst[] = {s->conc, s->thick, s->snow_thick, s->damage, s->ridge_ratio, s->sigma[0], s->sigma[1], s->sigma[2], s->conc_young, s->h_young, s->hs_young, s->conc_myi, s->thick_myi};
    double *src[] = {k->s.conc, k->s.thick, k->s.snow_thick, k->s.damage, k->s.ridge_ratio, k->s.sigma[0], k->s.sigma[1], k->s.sigma[2], k->s.conc_young, k->s.h_young, k->s.hs_young, k->s.conc_myi, k->s.thick_myi};
    for (size_t i = 0; i < sizeof dst / sizeof dst[0]; i++) memcpy(dst[i], src[i], Ne * D);
}

/* the lock-step phases of mr_thread with the spin barrier */
#define MRC_EACH(stmt) do { for (int r = tid; r < J->nranks; r += J->nthreads) { stmt; } mr_spin_barrier(c, &sense); } while (0)
static void mr_run_steps(ref_mr_ctx *c, int tid, int nsteps) {
    ref_mr_job *J = &c->job;
    const nxs_dyn_params *p = &c->p;
    const int S = p->substeps;
    const double dte = p->dtime_step / (double)S;
    int sense = atomic_load_explicit(&c->bar_sense, memory_order_acquire);
    for (int it = 0; it < nsteps; it++) {
        if (p->dynamics_type == NXS_DYN_NO_MOTION) break;
        if (p->dynamics_type == NXS_DYN_FREE_DRIFT) { MRC_EACH(ref_free_drift(J->m[r], p, J->s[r], J->f[r])); continue; }
        MRC_EACH(ref_prep(J->m[r], p, J->s[r], J->f[r], J->w[r]));
        for (int ss = 0; ss < S; ss++) {
            MRC_EACH(ref_substep_solve(J->m[r], p, J->s[r], J->f[r], J->w[r]); mr_pack(J, r));
            if (p->dynamics_type != NXS_DYN_MEVP) MRC_EACH(mr_unpack(J, r); ref_move_mesh(J->m[r], J->s[r], J->w[r], dte));
            else MRC_EACH(mr_unpack(J, r));
        }
        if (p->dynamics_type == NXS_DYN_MEVP) MRC_EACH(ref_move_mesh(J->m[r], J->s[r], J->w[r], p->dtime_step));
        for (int nit = 0; nit < 50; nit++) {  /* Q9 */
            MRC_EACH(ref_smoother_sweep(J->m[r], J->s[r], J->w[r]); mr_pack(J, r));
            MRC_EACH(mr_unpack(J, r));
        }
        MRC_EACH(ref_ow_tail(J->m[r], p, J->s[r], J->f[r], J->w[r]); ref_update(J->m[r], p, J->s[r], J->w[r]));
    }
}

static void *mr_ctx_thread(void *arg_) {
    ref_mr_targ *a = (ref_mr_targ *)arg_;
    ref_mr_ctx *c = a->c;
    const int tid = a->tid;
    free(a);
    if (c->cpu_of[tid] >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set); CPU_SET(c->cpu_of[tid], &set);
        if (pthread_setaffinity_np(pthread_self(), sizeof set, &set) != 0) c->cpu_of[tid] = -1;
    }
    int seen = 0;
    for (;;) {
        pthread_mutex_lock(&c->mu);
        while (c->generation == seen) pthread_cond_wait(&c->cv_go, &c->mu);
        seen = c->generation;
        const int cmd = c->command, nsteps = c->nsteps;
        pthread_mutex_unlock(&c->mu);
        int bad = 0;
        if (cmd == 5) {   /* ref_mr_configure: pinned to the planned CPU, or free on the process's mask */
            cpu_set_t set;
            CPU_ZERO(&set);
            if (c->pin && c->cpu_plan[tid] >= 0) CPU_SET(c->cpu_plan[tid], &set); else set = c->all_cpus;
            c->cpu_of[tid] = (pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0 && c->pin) ? c->cpu_plan[tid] : -1;
        }
        if (cmd == 1) { for (int r = tid; r < c->nranks; r += c->nthreads) bad |= mr_copy_in(c, r); }
        else if (cmd == 2) mr_run_steps(c, tid, nsteps);
        else if (cmd == 3) { for (int r = tid; r < c->nranks; r += c->nthreads) mr_copy_back(c, r); }
        pthread_mutex_lock(&c->mu);
        if (bad) c->failed = 1;
        if (++c->done == c->nthreads) pthread_cond_signal(&c->cv_done);
        pthread_mutex_unlock(&c->mu);
        if (cmd == 4) return NULL;
    }
}

static void mr_command(ref_mr_ctx *c, int cmd, int nsteps) {
    pthread_mutex_lock(&c->mu);
    c->command = cmd; c->nsteps = nsteps; c->done = 0; c->generation++;
    pthread_cond_broadcast(&c->cv_go);
    while (c->done < c->nthreads) pthread_cond_wait(&c->cv_done, &c->mu);
    pthread_mutex_unlock(&c->mu);
}

/* The CPUs this process may run on, ordered for pinning: physical cores (the first CPU of every thread_siblings_list) before their SMT siblings,
 * and inside each class the sockets in turn -- n threads then sit on n different cores spread over all the sockets.  Returns the count. */
static int mr_cpu_order(int *out, int cap, int *socket_of_out) {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return 0;
    enum { MAXC = CPU_SETSIZE };
    static int pkg[MAXC], primary[MAXC];
    int maxpkg = 0, n = 0;
    for (int cpu = 0; cpu < MAXC; cpu++) {
        pkg[cpu] = -1; primary[cpu] = 1;
        if (!CPU_ISSET(cpu, &set)) continue;
        char path[128];
        int v = 0;
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu);
        FILE *f = fopen(path, "r");
        if (f) { if (fscanf(f, "%d", &v) != 1) v = 0; fclose(f); }
        pkg[cpu] = v < 0 ? 0 : v;
        if (pkg[cpu] > maxpkg) maxpkg = pkg[cpu];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        f = fopen(path, "r");
        if (f) { int first = cpu; if (fscanf(f, "%d", &first) == 1) primary[cpu] = (first == cpu); fclose(f); }
    }
    for (int cls = 1; cls >= 0; cls--) {            /* primary siblings first */
        int cursor[64];
        for (int s = 0; s <= maxpkg && s < 64; s++) cursor[s] = 0;
        for (;;) {
            int added = 0;
            for (int s = 0; s <= maxpkg && s < 64; s++) {   /* one CPU of every socket per turn */
                int cpu = cursor[s];
                while (cpu < MAXC && !(pkg[cpu] == s && primary[cpu] == cls)) cpu++;
                cursor[s] = cpu + 1;
                if (cpu < MAXC && n < cap) { out[n] = cpu; if (socket_of_out) socket_of_out[n] = s; n++; added = 1; }
            }
            if (!added) break;
        }
    }
    return n;
}

ref_mr_ctx *ref_mr_create(int nranks, const nxs_dyn_mesh *const *m, const nxs_dyn_params *p, nxs_dyn_state *const *s, const nxs_dyn_forcing *const *f,
                          const nxs_dyn_halo *const *h, int nthreads, int pin) {
    if (nranks < 1 || !m || !p || !s || !f || !h) return NULL;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nranks) nthreads = nranks;
    ref_mr_ctx *c = (ref_mr_ctx *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->nranks = nranks; c->nthreads = nthreads; c->pin = pin; c->p = *p;
    c->src_m = m; c->src_s = s; c->src_f = f; c->src_h = h;
    c->rk = (ref_mr_rank *)calloc((size_t)nranks, sizeof(ref_mr_rank));
    c->pm = (const nxs_dyn_mesh **)calloc((size_t)nranks, sizeof(void *)); c->ps = (nxs_dyn_state **)calloc((size_t)nranks, sizeof(void *));
    c->pf = (const nxs_dyn_forcing **)calloc((size_t)nranks, sizeof(void *)); c->pw = (ref_work **)calloc((size_t)nranks, sizeof(void *));
    c->ph = (const nxs_dyn_halo **)calloc((size_t)nranks, sizeof(void *));
    c->th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    c->cpu_of = (int *)malloc((size_t)nthreads * sizeof(int));
    c->cpu_plan = (int *)malloc((size_t)nthreads * sizeof(int));
    CPU_ZERO(&c->all_cpus);
    if (sched_getaffinity(0, sizeof c->all_cpus, &c->all_cpus) != 0) for (int i = 0; i < CPU_SETSIZE; i++) CPU_SET(i, &c->all_cpus);
    ref_mr_job *J = &c->job;
    J->nranks = nranks; J->nthreads = nthreads;
    J->sendbuf = (double **)calloc((size_t)nranks, sizeof(double *));
    J->peer_seg = (int **)calloc((size_t)nranks, sizeof(int *));
    int ok = c->rk && c->pm && c->ps && c->pf && c->pw && c->ph && c->th && c->cpu_of && c->cpu_plan && J->sendbuf && J->peer_seg;
    for (int r = 0; r < nranks && ok; r++) {   /* who sends what to whom: checked on the caller's lists, as in ref_multirank_steps */
        const nxs_dyn_halo *hr = h[r];
        const int nr = hr->num_recv_procs;
        J->peer_seg[r] = (int *)malloc(sizeof(int) * (size_t)(nr + 1));
        if (!J->peer_seg[r]) { ok = 0; break; }
        for (int k = 0; k < nr && ok; k++) {
            const int q = hr->recv_procs[k];
            if (q < 0 || q >= nranks) { ok = 0; break; }
            int found = -1;
            for (int kk = 0; kk < h[q]->num_send_procs; kk++) if (h[q]->send_procs[kk] == hr->rank) found = kk;
            if (found < 0 || h[q]->send_offsets[found + 1] - h[q]->send_offsets[found] != hr->recv_offsets[k + 1] - hr->recv_offsets[k]) { ok = 0; break; }
            J->peer_seg[r][k] = found;
        }
    }
    if (ok) {
        for (int t = 0; t < nthreads; t++) { c->cpu_of[t] = -1; c->cpu_plan[t] = -1; }
        {
            int *order = (int *)malloc(sizeof(int) * CPU_SETSIZE), *sock = (int *)malloc(sizeof(int) * CPU_SETSIZE);
            const int n = (order && sock) ? mr_cpu_order(order, CPU_SETSIZE, sock) : 0;
            int seen_sock[64] = {0};
            for (int t = 0; t < nthreads && n > 0; t++) {
                c->cpu_plan[t] = order[t % n];
                if (pin) c->cpu_of[t] = c->cpu_plan[t];
                if (t < n && sock[t] >= 0 && sock[t] < 64 && !seen_sock[sock[t]]) { seen_sock[sock[t]] = 1; c->sockets_used++; }
            }
            free(order); free(sock);
        }
        pthread_barrier_init(&c->pbar, NULL, (unsigned)nthreads);
        pthread_mutex_init(&c->mu, NULL); pthread_cond_init(&c->cv_go, NULL); pthread_cond_init(&c->cv_done, NULL);
        atomic_init(&c->bar_count, 0); atomic_init(&c->bar_sense, 0);
        int started = 0;
        for (int t = 0; t < nthreads; t++) {
            ref_mr_targ *a = (ref_mr_targ *)malloc(sizeof *a);
            if (!a) break;
            a->c = c; a->tid = t;
            if (pthread_create(&c->th[t], NULL, mr_ctx_thread, a) != 0) { free(a); break; }
            started++;
        }
        if (started != nthreads) {   /* cannot run with fewer: end the ones that started */
            c->nthreads = started;
            if (started) { mr_command(c, 4, 0); for (int t = 0; t < started; t++) pthread_join(c->th[t], NULL); }
            c->nthreads = 0;
            ok = 0;
        }
    }
    if (ok) {
        mr_command(c, 1, 0);   /* every thread copies its partitions in: first touch */
        J->m = c->pm; J->p = &c->p; J->s = c->ps; J->f = c->pf; J->w = c->pw; J->h = c->ph;
        if (c->failed) ok = 0;
    }
    if (!ok) { ref_mr_destroy(c, 0); return NULL; }
    return c;
}

/* between runs: the barrier the phases meet at (0 spin, 1 sleeping) and whether the threads are pinned -- the same partitions, first touched as they were, so that
 * bench.py can time every combination on ONE context and report the best the host gives */
int ref_mr_configure(ref_mr_ctx *c, int barrier_kind, int pin) {
    if (!c || c->nthreads < 1 || barrier_kind < 0 || barrier_kind > 1) return -1;
    c->barrier_kind = barrier_kind;
    c->pin = pin ? 1 : 0;
    mr_command(c, 5, 0);
    return 0;
}

int ref_mr_run(ref_mr_ctx *c, int nsteps) {
    if (!c || nsteps < 0 || c->nthreads < 1) return -1;
    mr_command(c, 2, nsteps);
    return 0;
}

/* threads, CPUs pinned (-1 where not), sockets the pinned threads sit on */
int ref_mr_info(const ref_mr_ctx *c, int *nthreads, int *sockets_used, int *cpus, int cap) {
    if (!c) return -1;
    if (nthreads) *nthreads = c->nthreads;
    if (sockets_used) *sockets_used = c->sockets_used;
    for (int t = 0; cpus && t < c->nthreads && t < cap; t++) cpus[t] = c->cpu_of[t];
    return 0;
}

void ref_mr_destroy(ref_mr_ctx *c, int copy_back) {
    if (!c) return;
    if (c->nthreads > 0 && c->th) {
        if (copy_back && !c->failed) mr_command(c, 3, 0);
        mr_command(c, 4, 0);
        for (int t = 0; t < c->nthreads; t++) pthread_join(c->th[t], NULL);
        pthread_mutex_destroy(&c->mu); pthread_cond_destroy(&c->cv_go); pthread_cond_destroy(&c->cv_done);
        pthread_barrier_destroy(&c->pbar);
    }
    for (int r = 0; c->rk && r < c->nranks; r++) {
        for (int i = 0; i < c->rk[r].n_owned; i++) free(c->rk[r].owned[i]);
        free(c->rk[r].owned);
        ref_work_destroy(c->rk[r].w);
        if (c->job.sendbuf) free(c->job.sendbuf[r]);
        if (c->job.peer_seg) free(c->job.peer_seg[r]);
    }
    free(c->job.sendbuf); free(c->job.peer_seg);
    free(c->rk); free(c->pm); free(c->ps); free(c->pf); free(c->pw); free(c->ph); free(c->th); free(c->cpu_of); free(c->cpu_plan);
    free(c);
}
