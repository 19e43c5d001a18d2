// tests/native/patchcut_host.cpp -- TEST INFRASTRUCTURE.  Host build of the library's host-only mesh preparation (nextsim_amd/csrc/nxs_patchcut.hpp:
// patch cutter, Hilbert re-cut, D-ring patches, smoother patches, tables of the in-kernel halo exchange and of the resident loop;
// nextsim_amd/csrc/nxs_hull.inl: bamg's convex completion) with doors for tests/sanitize_worker.py.  tests/test_sanitizers.py compiles
// this file with -fsanitize=address,undefined and drives it through the inputs nxs_dyn_set_mesh meets: a numbering without locality, a
// large mesh followed by a small one, ragged partitions, every part of both BASELINE meshes, patches closed at 480 elements.  Every
// door also CHECKS what was built (each own node solved by one patch, each element written by one, fans complete and ascending, slots
// inside their rows ...), so a wrong table is caught here and not as a memory fault on a GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../nextsim_amd/csrc/nxs_guard.hpp"
#include "../../nextsim_amd/csrc/nxs_patchcut.hpp"
#include "../../nextsim_amd/csrc/nxs_hull.inl"

namespace {

using namespace nxs_cut;

struct Fail { std::string what; };
#define REQUIRE(cond, ...) do { if (!(cond)) { char b_[400]; snprintf(b_, sizeof b_, __VA_ARGS__); throw Fail{std::string(#cond) + ": " + b_}; } } while (0)

struct Mesh {
    std::vector<int> t[3];
    std::vector<unsigned char> ghost3;
    std::vector<double> x, y;
    int Nn = 0, Ne = 0, No = 0;
    MeshView view() const { return MeshView{t, ghost3.data(), x.data(), y.data(), Nn, Ne, No}; }
};

Mesh make_mesh(const int32_t *indices, const uint8_t *ghost3, const double *x, const double *y, int Nn, int Ne, int No) {
    Mesh m;
    m.Nn = Nn; m.Ne = Ne; m.No = No;
    for (int k = 0; k < 3; ++k) { m.t[k].resize(Ne); for (int e = 0; e < Ne; ++e) m.t[k][e] = indices[3 * e + k] - 1; }
    m.ghost3.assign(ghost3, ghost3 + 3 * (size_t)Ne);
    m.x.assign(x, x + Nn); m.y.assign(y, y + Nn);
    return m;
}

// the invariants of single-ring patches (what k_substep_fused / k_substep_resident index with)
void check_patches(const Mesh &m, const HostPatches &hp) {
    const int nP = hp.nP;
    REQUIRE(nP >= 1 || m.No == 0, "nP=%d", nP);
    REQUIRE(hp.Emax % 2 == 0 && hp.Mmax % 2 == 0 && hp.Pmax >= 1 && hp.Wp >= 1, "Emax=%d Mmax=%d Pmax=%d Wp=%d", hp.Emax, hp.Mmax, hp.Pmax, hp.Wp);
    REQUIRE((int)hp.own_cnt.size() == nP && (int)hp.elem_cnt.size() == nP && (int)hp.node_cnt.size() == nP, "count arrays");
    REQUIRE(hp.pnodes.size() == (size_t)nP * hp.Mmax && hp.pelem.size() == (size_t)nP * hp.Emax && hp.ptri.size() == (size_t)nP * hp.Emax * 4 &&
            hp.pfan.size() == (size_t)nP * hp.Wp * hp.Pmax, "array sizes");
    std::vector<int> off, adj;
    node_fans(m.t, m.Nn, m.Ne, off, adj);
    std::vector<int> solved(m.Nn, 0), written(m.Ne, 0), in_patch(m.Ne, -1);
    for (int q = 0; q < nP; ++q) {
        const int nO = hp.own_cnt[q], nE = hp.elem_cnt[q], nM = hp.node_cnt[q];
        REQUIRE(nO >= 0 && nO <= hp.Pmax && nE >= 0 && nE <= hp.Emax && nM >= nO && nM <= hp.Mmax, "patch %d: nO=%d nE=%d nM=%d", q, nO, nE, nM);
        const int *pn = hp.pnodes.data() + (size_t)q * hp.Mmax;
        for (int i = 0; i < nM; ++i) REQUIRE(pn[i] >= 0 && pn[i] < m.Nn, "patch %d slot %d names node %d", q, i, pn[i]);
        for (int i = 0; i < nO; ++i) { REQUIRE(pn[i] < m.No, "patch %d solves the ghost node %d", q, pn[i]); solved[pn[i]]++; }
        for (int l = 0; l < nE; ++l) {
            const int raw = hp.pelem[(size_t)q * hp.Emax + l], e = raw >= 0 ? raw : ~raw;
            REQUIRE(e >= 0 && e < m.Ne, "patch %d element slot %d names %d", q, l, e);
            if (l > 0) { const int rp = hp.pelem[(size_t)q * hp.Emax + l - 1]; REQUIRE((rp >= 0 ? rp : ~rp) < e, "patch %d: elements not ascending at slot %d", q, l); }
            if (raw >= 0) written[e]++;
            in_patch[e] = q;
            for (int k = 0; k < 3; ++k) {
                const int sl = hp.ptri[((size_t)q * hp.Emax + l) * 4 + k];
                REQUIRE(sl < nM && pn[sl] == m.t[k][e], "patch %d element %d corner %d: slot %d", q, e, k, sl);
                REQUIRE(sl < 1024, "corner slot %d does not fit ten bits", sl);
            }
        }
        // fans: exactly the node's elements, ascending, with the corner and its ghost flag
        for (int i = 0; i < nO; ++i) {
            const int n = pn[i];
            int k = 0;
            for (int j = off[n]; j < off[n + 1]; ++j, ++k) {
                REQUIRE(k < hp.Wp, "patch %d node %d: fan longer than Wp=%d", q, n, hp.Wp);
                const unsigned ent = hp.pfan[(size_t)q * hp.Wp * hp.Pmax + (size_t)k * hp.Pmax + i];
                REQUIRE(ent != 0xFFFFu, "patch %d node %d: fan entry %d missing", q, n, k);
                const int l = (int)(ent >> 3), c = (int)(ent & 3u), e = adj[j];
                REQUIRE(l < nE, "patch %d node %d: fan names element slot %d of %d", q, n, l, nE);
                const int raw = hp.pelem[(size_t)q * hp.Emax + l];
                REQUIRE((raw >= 0 ? raw : ~raw) == e && c < 3 && m.t[c][e] == n, "patch %d node %d: fan entry %d is not element %d", q, n, k, e);
                REQUIRE(((ent & 4u) != 0) == (m.ghost3[3 * (size_t)e + c] != 0), "patch %d node %d: ghost flag of element %d", q, n, e);
            }
            for (; k < hp.Wp; ++k) REQUIRE(hp.pfan[(size_t)q * hp.Wp * hp.Pmax + (size_t)k * hp.Pmax + i] == 0xFFFFu, "patch %d node %d: stale fan entry %d", q, n, k);
        }
    }
    for (int n = 0; n < m.No; ++n) REQUIRE(solved[n] == 1, "own node %d is solved by %d patches", n, solved[n]);
    for (int e = 0; e < m.Ne; ++e) REQUIRE(written[e] == 1, "element %d is written by %d patches", e, written[e]);
}

void check_patches2(const Mesh &m, const HostPatches2 &hp) {
    const int nP = hp.nP, D = hp.D;
    std::vector<int> off, adj;
    node_fans(m.t, m.Nn, m.Ne, off, adj);
    std::vector<int> written(m.Ne, 0), own(m.Nn, 0);
    REQUIRE(hp.pnodes.size() == (size_t)nP * hp.NDmax && hp.pelem.size() == (size_t)nP * hp.EDmax && hp.pfan.size() == (size_t)nP * hp.Wp * hp.NSmax, "array sizes");
    for (int q = 0; q < nP; ++q) {
        const int *nc = hp.ncnt.data() + (size_t)q * (D + 1), *ec = hp.ecnt.data() + (size_t)q * D;
        for (int l = 0; l < D; ++l) REQUIRE(nc[l] <= nc[l + 1] && (l == 0 || ec[l - 1] <= ec[l]), "patch %d: levels not nested", q);
        REQUIRE(nc[D] <= hp.NDmax && nc[D - 1] <= hp.NSmax && ec[D - 1] <= hp.EDmax, "patch %d: level sizes", q);
        const int *pn = hp.pnodes.data() + (size_t)q * hp.NDmax;
        for (int i = 0; i < nc[0]; ++i) own[pn[i]]++;
        for (int l = 0; l < ec[D - 1]; ++l) {
            const int raw = hp.pelem[(size_t)q * hp.EDmax + l], e = raw >= 0 ? raw : ~raw;
            REQUIRE(e >= 0 && e < m.Ne, "patch %d: element %d", q, e);
            if (raw >= 0) written[e]++;
            for (int k = 0; k < 3; ++k) {
                const int sl = hp.ptri[((size_t)q * hp.EDmax + l) * 4 + k];
                REQUIRE(sl < nc[D] && pn[sl] == m.t[k][e], "patch %d element %d corner %d", q, e, k);
            }
        }
        for (int i = 0; i < nc[D - 1]; ++i) {  // every solved node has its complete fan among the patch's elements
            const int n = pn[i];
            int k = 0;
            for (int j = off[n]; j < off[n + 1]; ++j, ++k) {
                const unsigned ent = hp.pfan[(size_t)q * hp.Wp * hp.NSmax + (size_t)k * hp.NSmax + i];
                REQUIRE(ent != 0xFFFFu && (int)(ent >> 3) < ec[D - 1], "patch %d node %d: fan entry %d", q, n, k);
                const int raw = hp.pelem[(size_t)q * hp.EDmax + (ent >> 3)];
                REQUIRE((raw >= 0 ? raw : ~raw) == adj[j], "patch %d node %d: fan entry %d is not element %d", q, n, k, adj[j]);
            }
        }
    }
    for (int n = 0; n < m.Nn; ++n) REQUIRE(own[n] == 1, "node %d is an own node of %d patches", n, own[n]);
    for (int e = 0; e < m.Ne; ++e) REQUIRE(written[e] == 1, "element %d is written by %d patches", e, written[e]);
}

void put_msg(char *msg, int cap, const std::string &s) { if (msg && cap > 0) { snprintf(msg, (size_t)cap, "%s", s.c_str()); } }

int caught(char *msg, int cap) {
    try { throw; }
    catch (const Fail &f) { put_msg(msg, cap, f.what); return 1; }
    catch (...) { return nxs_guard::caught("patchcut_host", [&](int, const char *t) { put_msg(msg, cap, t); }); }
}

}  // namespace

static int g_band_nodes = 0;   // pc_set_band: the size of the patches along the partition boundary in the cuts that follow (0: like the others)

extern "C" {

void pc_set_band(int band_nodes) { g_band_nodes = band_nodes; }

// One rank's mesh through everything nxs_dyn_set_mesh / set_halo / the first step build on the host.  stats (int64[20]): nP, Pmax, Emax,
// Mmax, Wp, used_hilbert, fused_lds, P, resident ok, resident max neighbours, n_boundary, reordered, multi nP, multi EDmax, smoother nP, smoother NDmax, cut for the large-patch resident kernel.
// Returns 0, 1 (a check failed: msg says which) or an NXS_ERR_* of the guard (an exception: msg says which).
int pc_run(const int32_t *indices, const uint8_t *ghost3, const double *x, const double *y, int Nn, int Ne, int No, int patch_nodes,
           int want_resident, int cus, int res_ept, int ns, const int32_t *send_offsets, const int32_t *send_index, int nr,
           const int32_t *recv_offsets, const int32_t *recv_index, int overlap, const int32_t *n2n /*[W2][Nn] or NULL*/, const int32_t *n2n_cnt,
           int W2, int depth_multi, int depth_smooth, int64_t *stats, char *msg, int msg_cap) try {
    put_msg(msg, msg_cap, "");
    for (int i = 0; i < 20; ++i) stats[i] = 0;
    const Mesh m = make_mesh(indices, ghost3, x, y, Nn, Ne, No);
    PatchPlan plan;
    std::vector<char> sent((size_t)std::max(No, 1), 0);   // as nxs_dyn_set_halo hands it to the cutter: the own nodes this rank sends
    for (int j = 0; send_index && ns > 0 && j < send_offsets[ns]; ++j) if (send_index[j] >= 0 && send_index[j] < No) sent[send_index[j]] = 1;
    MeshView mv = m.view();
    if (ns > 0) mv.sent = sent.data();
    const std::string why = plan_patches(mv, patch_nodes, want_resident != 0, cus, plan, res_ept, true, g_band_nodes);
    if (!why.empty()) { put_msg(msg, msg_cap, why); return 2; }
    HostPatches &hp = plan.hp;
    check_patches(m, hp);
    REQUIRE(plan.fused_lds == fused_lds_of(hp) && plan.fused_lds <= 160 * 1024, "fused_lds=%zu", plan.fused_lds);
    std::vector<int> pet;
    pack_pet(hp, pet);
    REQUIRE(pet.size() == 2 * (size_t)hp.nP * hp.Emax, "pet size");
    stats[16] = plan.cut_big; for (int q = 0; q < hp.nP; ++q) { stats[17] += hp.elem_cnt[q]; stats[18] += (hp.elem_cnt[q] + 63) / 64 > 24; stats[19] = std::max<int64_t>(stats[19], hp.node_cnt[q] - hp.own_cnt[q]); } stats[0] = hp.nP; stats[1] = hp.Pmax; stats[2] = hp.Emax; stats[3] = hp.Mmax; stats[4] = hp.Wp; stats[5] = hp.used_hilbert; stats[6] = (int64_t)plan.fused_lds; stats[7] = plan.P;
    const bool mr = ns > 0 || nr > 0 || No < Nn;
    if (mr && g_band_nodes > 0 && want_resident && !plan.cut_big && g_band_nodes < plan.P) {
        // the own nodes that share an element with a ghost (on the partitions a mesh partitioner makes: the nodes this rank sends; a ragged element
        // partition also sends nodes whose neighbours' elements it does not hold) sit in patches of their own, of at most g_band_nodes nodes
        std::vector<char> band(No, 0);
        for (int e = 0; e < Ne; ++e) {
            bool has_ghost = false;
            for (int k = 0; k < 3; ++k) has_ghost = has_ghost || indices[3 * e + k] - 1 >= No;
            if (has_ghost) for (int k = 0; k < 3; ++k) if (indices[3 * e + k] - 1 < No) band[indices[3 * e + k] - 1] = 1;
        }
        if (ns > 0) for (int n = 0; n < No; ++n) if (sent[n]) band[n] = 1;   // (round 5: and every node this rank sends, ghost nearby or not)
        for (int q = 0; q < hp.nP; ++q) {
            int nb = 0;
            for (int i = 0; i < hp.own_cnt[q]; ++i) nb += band[hp.pnodes[(size_t)q * hp.Mmax + i]];
            REQUIRE(nb == 0 || (nb == hp.own_cnt[q] && nb <= g_band_nodes), "patch %d mixes %d boundary nodes with %d others", q, nb, hp.own_cnt[q] - nb);
            // hence: a patch that takes part in the exchange between ranks (it sends, or it stages a ghost) is one of the SMALL ones
            bool exch = false;
            for (int i = 0; i < hp.node_cnt[q]; ++i) { const int n = hp.pnodes[(size_t)q * hp.Mmax + i]; exch = exch || n >= No || (i < hp.own_cnt[q] && ns > 0 && sent[n]); }
            REQUIRE(!exch || hp.own_cnt[q] == 0 || hp.own_cnt[q] <= g_band_nodes, "patch %d of %d own nodes takes part in the exchange", q, hp.own_cnt[q]);
        }
    }
    if (!mr) {  // the rows of k_prep_fused: every node's elements in an order of its own (descending here; bamg's is a chained list's) -> patch slots
        std::vector<std::vector<int>> fan(Nn);
        for (int e = Ne - 1; e >= 0; --e) for (int k = 0; k < 3; ++k) fan[indices[3 * e + k] - 1].push_back(e);
        int W1 = 1;
        for (const auto &f : fan) W1 = std::max(W1, (int)f.size());
        W1 += 1;  // (a pad column, as bamg's table has NaNs)
        std::vector<int> n2e((size_t)W1 * Nn, -1);
        for (int n = 0; n < Nn; ++n) for (size_t j = 0; j < fan[n].size(); ++j) n2e[j * Nn + n] = fan[n][j];
        std::vector<unsigned short> rows;
        REQUIRE(build_prep_rows(hp, n2e.data(), W1, Nn, rows), "build_prep_rows refused a table of this mesh%s", "");
        REQUIRE(rows.size() == (size_t)hp.nP * W1 * hp.Pmax, "rows size");
        for (int q = 0; q < hp.nP; ++q)
            for (int i = 0; i < hp.own_cnt[q]; ++i) {
                const int n = hp.pnodes[(size_t)q * hp.Mmax + i];
                for (int j = 0; j < W1; ++j) {
                    const unsigned short sl = rows[((size_t)q * W1 + j) * hp.Pmax + i];
                    const int want = n2e[(size_t)j * Nn + n];
                    if (want < 0) { REQUIRE(sl == 0xFFFF, "patch %d node %d row %d: a slot for a pad", q, n, j); continue; }
                    REQUIRE(sl < hp.elem_cnt[q], "patch %d node %d row %d: slot %d", q, n, j, (int)sl);
                    const int raw = hp.pelem[(size_t)q * hp.Emax + sl];
                    REQUIRE((raw >= 0 ? raw : ~raw) == want, "patch %d node %d row %d: element %d, not %d", q, n, j, raw >= 0 ? raw : ~raw, want);
                }
            }
        REQUIRE(prep_fused_lds_of(hp) >= 48 * (size_t)hp.Emax, "prep lds");
        if (Ne >= 2 && hp.nP >= 1 && hp.own_cnt[0] >= 1) {  // a row that names an element that does not touch the node: refused, not mapped
            const int n0 = hp.pnodes[0];
            int far = -1;
            for (int e = 0; e < Ne && far < 0; ++e) {
                bool in0 = false;
                for (int l = 0; l < hp.elem_cnt[0]; ++l) { const int raw = hp.pelem[l]; in0 = in0 || (raw >= 0 ? raw : ~raw) == e; }
                if (!in0) far = e;
            }
            if (far >= 0) { n2e[(size_t)(W1 - 1) * Nn + n0] = far; REQUIRE(!build_prep_rows(hp, n2e.data(), W1, Nn, rows), "a foreign element was accepted%s", ""); }
        }
    }
    if (mr) {
        const std::vector<int> so(send_offsets, send_offsets + ns + 1), ro(recv_offsets, recv_offsets + nr + 1);
        const std::vector<int> si(send_index, send_index + so[ns]), ri(recv_index, recv_index + ro[nr]);
        const HaloLists hl{&so, &ro, &si, &ri, ns, nr};
        HaloFusedPlan hf;
        const std::string w2 = plan_halo_fused(Nn, No, hl, hp, hf);
        if (!w2.empty()) { put_msg(msg, msg_cap, w2); return 2; }
        check_patches(m, hp);  // (rewritten boundary-first)
        // boundary patches lead; every sent node has its (neighbour, position) entries; every ghost its mailbox slot
        REQUIRE(hf.n_boundary >= 0 && hf.n_boundary <= hp.nP, "n_boundary=%d", hf.n_boundary);
        for (int q = 0; q < hp.nP; ++q) {
            bool bnd = false;
            for (int i = 0; i < hp.node_cnt[q]; ++i) {
                const int n = hp.pnodes[(size_t)q * hp.Mmax + i];
                bnd = bnd || n >= No || (i < hp.own_cnt[q] && hf.sptr[n + 1] > hf.sptr[n]);
            }
            REQUIRE(bnd == (q < hf.n_boundary), "patch %d: boundary=%d but n_boundary=%d", q, (int)bnd, hf.n_boundary);
        }
        REQUIRE(hf.sptr[No] == so[ns], "%d send entries for %d sent nodes", hf.sptr[No], so[ns]);
        for (int n = 0; n < No; ++n)
            for (int j = hf.sptr[n]; j < hf.sptr[n + 1]; ++j) {
                const int k = hf.sk[j], pos = hf.spos[j];
                REQUIRE(k >= 0 && k < ns && pos >= 0 && pos < so[k + 1] - so[k] && si[so[k] + pos] == n, "send entry %d of node %d", j, n);
            }
        for (int g = 0; g < Nn - No; ++g) {
            const int k = hf.gk[g];
            REQUIRE(k >= 0 && k < nr && hf.gsrl[g] == ro[k + 1] - ro[k], "ghost %d: neighbour %d", g, k);
            const int j = hf.goff[g] - 2 * ro[k];
            REQUIRE(j >= 0 && j < hf.gsrl[g] && ri[ro[k] + j] == No + g, "ghost %d: mailbox offset %d", g, hf.goff[g]);
        }
        stats[10] = hf.n_boundary; stats[11] = hf.reordered;
    }
    {
        ResidentPlan rp;
        const bool big = resident_is_big(hp), ovl = overlap != 0 && (mr || big);
        plan_resident(hp, Nn, No, mr, ns, ovl, rp);
        stats[8] = rp.ok; stats[9] = rp.max_nbr;
        if (rp.ok) {
            for (int q = 0; q < hp.nP; ++q) {
                REQUIRE(rp.cnt[q] >= 0 && rp.cnt[q] <= NXS_CUT_RES_NBR, "patch %d waits for %d patches", q, rp.cnt[q]);
                for (int k = 0; k < rp.cnt[q]; ++k) { const int o = rp.nbr[(size_t)q * NXS_CUT_RES_NBR + k]; REQUIRE(o >= 0 && o < hp.nP && o != q, "patch %d neighbour %d", q, o); }
            }
            if (ovl) {  // the overlap variant's lists are a permutation of the patch's with interior elements first
                HostPatches alt = hp;
                alt.pelem = rp.rpelem; alt.ptri = rp.rptri; alt.pfan = rp.rpfan;
                for (int q = 0; q < hp.nP; ++q) {
                    REQUIRE(rp.ecut[q] % (big ? 512 : 64) == 0 && rp.ecut[q] <= hp.elem_cnt[q], "patch %d: ecut=%d", q, rp.ecut[q]);
                    std::vector<int> a(hp.pelem.begin() + (size_t)q * hp.Emax, hp.pelem.begin() + (size_t)q * hp.Emax + hp.elem_cnt[q]);
                    std::vector<int> b(rp.rpelem.begin() + (size_t)q * hp.Emax, rp.rpelem.begin() + (size_t)q * hp.Emax + hp.elem_cnt[q]);
                    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
                    REQUIRE(a == b, "patch %d: the overlap list is not a permutation", q);
                    for (int l = 0; l < rp.ecut[q]; ++l)
                        for (int k = 0; k < 3; ++k) REQUIRE(rp.rptri[((size_t)q * hp.Emax + l) * 4 + k] < hp.own_cnt[q], "patch %d: early element %d touches a halo node", q, l);
                }
            }
        }
    }
    if (!mr && n2n && n2n_cnt) {
        const std::vector<int> vn(n2n, n2n + (size_t)W2 * Nn), vc(n2n_cnt, n2n_cnt + Nn);
        if (depth_multi >= 2) {
            Patch2Plan p2;
            const std::string w3 = plan_patches2(m.view(), hp.used_hilbert, 0, depth_multi, false, cus, vn, vc, W2, p2);
            if (w3.empty()) {
                check_patches2(m, p2.hp);
                REQUIRE(p2.lds <= 160 * 1024, "multi lds %zu", p2.lds);
                stats[12] = p2.hp.nP; stats[13] = p2.hp.EDmax;
            }
        }
        {   // the same mesh cut for k_substep_pair (depth 2, two workgroups per CU): the largest patches that fit, or one full round of smaller ones
            Patch2Plan pk;
            const std::string w5 = plan_patches2(m.view(), hp.used_hilbert, 0, 2, false, cus, vn, vc, W2, pk, true);
            if (w5.empty()) {
                check_patches2(m, pk.hp);
                int own_max = 0;
                for (int q = 0; q < pk.hp.nP; ++q) own_max = std::max(own_max, pk.hp.ncnt[(size_t)q * 3]);
                REQUIRE(pk.pair_kernel && pk.threads == 512 && pk.lds == pair_lds_of(pk.hp) && pk.lds <= 80 * 1024 && pair_kernel_fits(pk.hp, own_max),
                        "pair plan: P=%d lds=%zu EDmax=%d ESmax=%d NSmax=%d own=%d", pk.P, pk.lds, pk.hp.EDmax, pk.hp.ESmax, pk.hp.NSmax, own_max);
                // the size is the largest that fits, or smaller for the sake of the rounds of workgroups: one WHOLE round, or -- more than a round --
                // just under a multiple of half a round (one workgroup per CU)
                REQUIRE(pk.P_fit >= pk.P && pk.P >= 64, "pair plan: P=%d P_fit=%d", pk.P, pk.P_fit);
                const int slots = 2 * std::max(cus, 1), unit = std::max(cus, 1), hr = (pk.hp.nP + unit - 1) / unit;
                if (pk.P < pk.P_fit && pk.hp.nP > slots)
                    REQUIRE(pk.hp.nP <= hr * unit && (double)pk.hp.nP >= 0.9 * hr * unit - 1., "pair plan: %d patches of %d nodes (largest that fit: %d) on %d slots", pk.hp.nP, pk.P, pk.P_fit, slots);
            }
            if (w5.empty()) {   // the ready-made fan indices of k_substep_pair / k_substep_multi: every entry names a corner force of the patch or the pair of zeros
                const HostPatches2 &x = pk.hp;
                std::vector<unsigned int> f8;
                decode_fan8(x, f8);
                REQUIRE(f8.size() == (size_t)x.nP * 4 * x.NSmax, "decode_fan8 size%s", "");
                const unsigned zidx = 3u * (unsigned)x.EDmax;
                for (int q = 0; q < x.nP; ++q)
                    for (int i = 0; i < x.ncnt[(size_t)q * 3 + 1]; ++i)
                        for (int k = 0; k < 8; ++k) {
                            const unsigned idx = (f8[((size_t)q * 4 + k / 2) * x.NSmax + i] >> (16 * (k & 1))) & 0xFFFFu;
                            const unsigned ent = k < x.Wp ? x.pfan[((size_t)q * x.Wp + k) * x.NSmax + i] : 0xFFFFu;
                            if (ent == 0xFFFFu || (ent & 4u)) REQUIRE(idx == zidx, "patch %d node %d entry %d: a pad or ghost corner must name the pair of zeros", q, i, k);
                            else REQUIRE(idx == (ent & 3u) * (unsigned)x.EDmax + (ent >> 3) && (ent >> 3) < (unsigned)x.ecnt[(size_t)q * 2 + 1] && (ent & 3u) < 3u,
                                         "patch %d node %d entry %d: index %u", q, i, k, idx);
                        }
            }
            if (w5.empty()) {   // k_substep_flow: what a patch waits for -- the writers of its outer elements, the owners of its staged nodes, the reverse relation, itself
                const HostPatches2 &x = pk.hp;
                std::vector<int> ptr, dep;
                REQUIRE(build_flow_deps(x, Nn, Ne, ptr, dep), "build_flow_deps refused a single-rank cut%s", "");
                REQUIRE((int)ptr.size() == x.nP + 1 && ptr[0] == 0 && ptr[x.nP] == (int)dep.size(), "flow deps: CSR shape%s", "");
                std::vector<int> owner(Nn, -1), writer(Ne, -1);
                for (int q = 0; q < x.nP; ++q) {
                    for (int i = 0; i < x.ncnt[(size_t)q * 3]; ++i) owner[x.pnodes[(size_t)q * x.NDmax + i]] = q;
                    for (int l = 0; l < x.ecnt[(size_t)q * 2 + 1]; ++l) if (x.pelem[(size_t)q * x.EDmax + l] >= 0) writer[x.pelem[(size_t)q * x.EDmax + l]] = q;
                }
                auto has = [&](int q, int r) { return std::binary_search(dep.begin() + ptr[q], dep.begin() + ptr[q + 1], r); };
                for (int q = 0; q < x.nP; ++q) {
                    REQUIRE(std::is_sorted(dep.begin() + ptr[q], dep.begin() + ptr[q + 1]) && std::adjacent_find(dep.begin() + ptr[q], dep.begin() + ptr[q + 1]) == dep.begin() + ptr[q + 1],
                            "flow deps of patch %d not strictly ascending", q);
                    REQUIRE(has(q, q), "patch %d does not wait for itself", q);
                    for (int i = 0; i < x.ncnt[(size_t)q * 3 + 2]; ++i) REQUIRE(has(q, owner[x.pnodes[(size_t)q * x.NDmax + i]]), "patch %d: the owner of staged node %d is missing", q, i);
                    for (int l = 0; l < x.ecnt[(size_t)q * 2 + 1]; ++l) {
                        int e = x.pelem[(size_t)q * x.EDmax + l];
                        if (e < 0) e = ~e;
                        REQUIRE(has(q, writer[e]), "patch %d: the writer of element slot %d is missing", q, l);
                    }
                    for (int j = ptr[q]; j < ptr[q + 1]; ++j) REQUIRE(dep[j] >= 0 && dep[j] < x.nP && has(dep[j], q), "flow deps not symmetric: %d -> %d", q, dep[j]);
                }
                int qs[9];
                flow_queues(x.nP, qs);
                REQUIRE(qs[0] == 0 && qs[8] == x.nP, "flow queues do not cover the patches%s", "");
                for (int k = 0; k < 8; ++k) REQUIRE(qs[k + 1] - qs[k] == x.nP / 8 + (k < x.nP % 8 ? 1 : 0), "flow queue %d has %d patches", k, qs[k + 1] - qs[k]);
                HostPatches2 broken = x;   // an element without a writer: refused
                bool cut = false;
                for (size_t i = 0; i < broken.pelem.size() && !cut; ++i) if (broken.pelem[i] >= 0 && (int)(i % x.EDmax) < broken.ecnt[(i / x.EDmax) * 2 + 1]) { broken.pelem[i] = ~broken.pelem[i]; cut = true; }
                REQUIRE(!cut || !build_flow_deps(broken, Nn, Ne, ptr, dep), "an element without a writer accepted%s", "");
            }
            if (w5.empty()) {   // after a regrid: the size kept before is tried first (kept if it fits; a hint that no longer fits is searched below)
                Patch2Plan again, above;
                REQUIRE(plan_patches2(m.view(), hp.used_hilbert, 0, 2, false, cus, vn, vc, W2, again, true, pk.P).empty() && again.P == pk.P && again.pair_kernel, "hint %d -> %d", pk.P, again.P);
                check_patches2(m, again.hp);
                REQUIRE(plan_patches2(m.view(), hp.used_hilbert, 0, 2, false, cus, vn, vc, W2, above, true, pk.P + 40).empty() && above.pair_kernel && above.lds <= 80 * 1024,
                        "hint %d -> %d", pk.P + 40, above.P);
                check_patches2(m, above.hp);
            }
            Patch2Plan bad;   // patches of 1 000 nodes cannot fit it: refused with a reason, never cut
            REQUIRE(Nn < 1000 || !plan_patches2(m.view(), hp.used_hilbert, 1000, 2, false, cus, vn, vc, W2, bad, true).empty(), "1 000-node patches accepted%s", "");
        }
        if (depth_smooth >= 1) {
            SmoothPlan sp;
            const std::string w4 = plan_smooth_patches(m.view(), hp.used_hilbert, depth_smooth, vn, vc, W2, sp);
            if (w4.empty()) {
                for (int q = 0; q < sp.nP; ++q) {
                    const int nS = sp.ncnt[(size_t)q * (sp.D + 1) + sp.D - 1], nD = sp.ncnt[(size_t)q * (sp.D + 1) + sp.D];
                    REQUIRE(nS <= sp.NSmax && nD <= sp.NDmax, "smoother patch %d", q);
                    for (int i = 0; i < nS; ++i) {
                        const int n = sp.pnodes[(size_t)q * sp.NDmax + i];
                        for (int k = 0; k < vc[n]; ++k) {
                            const int sl = sp.pnbr[((size_t)q * W2 + k) * sp.NSmax + i];
                            REQUIRE(sl < nD && sp.pnodes[(size_t)q * sp.NDmax + sl] == vn[(size_t)k * Nn + n], "smoother patch %d node %d neighbour %d", q, n, k);
                        }
                    }
                }
                stats[14] = sp.nP; stats[15] = sp.NDmax;
            }
        }
    }
    return 0;
} catch (...) { return caught(msg, msg_cap); }

// The two-ring patches of k_substep_pair<HALO> (several ranks): nxs_cut::plan_pair_patches_mr on this rank's mesh and send list, and every invariant the kernel
// and its exchange rely on.  out: [0] nP, [1] nG, [2] nBand, [3] P, [4] LDS bytes, [5] sum |E_2|, [6] sum |N_2|, [7] patches of elements without an own node.
int pc_pair_mr(const int32_t *indices, const uint8_t *ghost3, const double *x, const double *y, int Nn, int Ne, int No, int pair_nodes, int cus,
               const int32_t *send_index, int n_send, int hilbert, int64_t *out, char *msg, int msg_cap) try {
    put_msg(msg, msg_cap, "");
    const Mesh m = make_mesh(indices, ghost3, x, y, Nn, Ne, No);
    std::vector<char> sent((size_t)std::max(No, 1), 0);
    for (int i = 0; i < n_send; ++i) { REQUIRE(send_index[i] >= 0 && send_index[i] < No, "send_index[%d]", i); sent[send_index[i]] = 1; }
    PairHaloPlan plan;
    const std::string why = plan_pair_patches_mr(m.view(), hilbert != 0, pair_nodes, cus, sent, plan);
    if (!why.empty()) { put_msg(msg, msg_cap, why); return 1; }
    const HostPatches2 &hp = plan.hp;
    const int nP = hp.nP;
    REQUIRE(hp.D == 2 && (int)plan.pflags.size() == nP && plan.nBand <= plan.nG && plan.nG <= nP, "counts: nP=%d nG=%d nBand=%d", nP, plan.nG, plan.nBand);
    REQUIRE(plan.lds <= 80 * 1024 && hp.EDmax <= 3 * 512 && hp.ESmax <= 2 * 512 && hp.NSmax <= 2 * 512 && hp.NDmax <= 1024, "limits: lds=%zu EDmax=%d ESmax=%d NSmax=%d NDmax=%d", plan.lds, hp.EDmax, hp.ESmax, hp.NSmax, hp.NDmax);
    REQUIRE(hp.pnodes.size() == (size_t)nP * hp.NDmax && hp.pelem.size() == (size_t)nP * hp.EDmax && hp.ptri.size() == (size_t)nP * hp.EDmax * 4 && hp.pfan.size() == (size_t)nP * hp.Wp * hp.NSmax, "array sizes");
    std::vector<int> off, adj;
    node_fans(m.t, m.Nn, m.Ne, off, adj);
    std::vector<int> own(Nn, 0), written(Ne, 0), noted(Nn, 0), slot(Nn, -1);
    long long sumE2 = 0, sumN2 = 0;
    int orphan_patches = 0;
    for (int q = 0; q < nP; ++q) {
        const int *nc = hp.ncnt.data() + (size_t)q * 3, *ec = hp.ecnt.data() + (size_t)q * 2;
        const int *pn = hp.pnodes.data() + (size_t)q * hp.NDmax;
        const unsigned fl = plan.pflags[q];
        REQUIRE(nc[0] <= 512 && nc[0] <= nc[1] && nc[1] <= nc[2] && ec[0] <= ec[1] && nc[2] <= hp.NDmax && nc[1] <= hp.NSmax && ec[1] <= hp.EDmax && ec[0] <= hp.ESmax, "patch %d: levels", q);
        REQUIRE((q < plan.nBand) == ((fl & 6u) != 0) && (q < plan.nG) == ((fl & 1u) != 0), "patch %d: flags %u out of order (nBand=%d nG=%d)", q, fl, plan.nBand, plan.nG);
        if (nc[0] == 0) ++orphan_patches;
        sumE2 += ec[1]; sumN2 += nc[2];
        bool sends = false, ghost1 = false, ghost2 = false;
        for (int i = 0; i < nc[2]; ++i) {
            const int n = pn[i];
            REQUIRE(n >= 0 && n < Nn && slot[n] == -1, "patch %d slot %d: node %d (twice?)", q, i, n);
            slot[n] = i;
            if (i < nc[0]) { REQUIRE(n < No, "patch %d solves ghost %d as its own", q, n); own[n]++; sends = sends || sent[n]; }
            if (n >= No) { if (i < nc[1]) { ghost1 = true; noted[n]++; } ghost2 = true; }
        }
        REQUIRE(((fl & 2u) != 0) == sends && ((fl & 4u) != 0) == ghost1 && ((fl & 1u) != 0) == (ghost2 || sends), "patch %d: flags %u but sends=%d ghost in N_1=%d in N_2=%d", q, fl, sends, ghost1, ghost2);
        std::vector<char> in1(Ne, 0);   // (small meshes only: the test meshes)
        for (int l = 0; l < ec[1]; ++l) {
            const int raw = hp.pelem[(size_t)q * hp.EDmax + l], e = raw >= 0 ? raw : ~raw;
            REQUIRE(e >= 0 && e < Ne, "patch %d: element %d", q, e);
            if (l < ec[0]) { if (raw >= 0) written[e]++; in1[e] = 1; }
            else REQUIRE(true, "-");
            if (l > 0 && l != ec[0]) { const int rp = hp.pelem[(size_t)q * hp.EDmax + l - 1]; REQUIRE((rp >= 0 ? rp : ~rp) < e, "patch %d: level not ascending at slot %d", q, l); }
            bool touches_own_of_patch = false;
            for (int k = 0; k < 3; ++k) {
                const int sl = hp.ptri[((size_t)q * hp.EDmax + l) * 4 + k];
                REQUIRE(sl < nc[l < ec[0] ? 1 : 2] && pn[sl] == m.t[k][e], "patch %d element %d corner %d: slot %d", q, e, k, sl);
                touches_own_of_patch = touches_own_of_patch || sl < nc[0];
            }
            if (nc[0] > 0) REQUIRE((l < ec[0]) == touches_own_of_patch, "patch %d: element %d in the wrong level", q, e);
            else { bool any_own = false; for (int k = 0; k < 3; ++k) any_own = any_own || m.t[k][e] < No; REQUIRE(!any_own && l < ec[0], "patch %d (no own nodes) holds element %d with an own node", q, e); }
        }
        for (int i = 0; i < nc[1]; ++i) {   // own nodes of N_1: the complete fan, ascending, inside E_2 (E_1 for the patch's own nodes); ghosts: no row
            const int n = pn[i];
            int k = 0;
            if (n < No)
                for (int j = off[n]; j < off[n + 1]; ++j, ++k) {
                    REQUIRE(k < hp.Wp, "patch %d node %d: fan longer than Wp", q, n);
                    const unsigned ent = hp.pfan[(size_t)q * hp.Wp * hp.NSmax + (size_t)k * hp.NSmax + i];
                    REQUIRE(ent != 0xFFFFu && (int)(ent >> 3) < (i < nc[0] ? ec[0] : ec[1]), "patch %d node %d: fan entry %d", q, n, k);
                    const int raw = hp.pelem[(size_t)q * hp.EDmax + (ent >> 3)], e = adj[j], c = (int)(ent & 3u);
                    REQUIRE((raw >= 0 ? raw : ~raw) == e && c < 3 && m.t[c][e] == n, "patch %d node %d: fan entry %d is not element %d", q, n, k, e);
                    REQUIRE(((ent & 4u) != 0) == (m.ghost3[3 * (size_t)e + c] != 0), "patch %d node %d: ghost flag of element %d", q, n, e);
                }
            for (; k < hp.Wp; ++k) REQUIRE(hp.pfan[(size_t)q * hp.Wp * hp.NSmax + (size_t)k * hp.NSmax + i] == 0xFFFFu, "patch %d node %d: stale fan entry %d", q, n, k);
        }
        for (int i = 0; i < nc[0]; ++i)    // E_1 holds EVERY element of an own node
            for (int j = off[pn[i]]; j < off[pn[i] + 1]; ++j) REQUIRE(in1[adj[j]], "patch %d: element %d of own node %d is not in E_1", q, adj[j], pn[i]);
        for (int i = 0; i < nc[2]; ++i) slot[pn[i]] = -1;
    }
    for (int n = 0; n < No; ++n) REQUIRE(own[n] == 1, "own node %d is solved by %d patches", n, own[n]);
    for (int n = No; n < Nn; ++n) REQUIRE(noted[n] >= 1, "ghost node %d is in no patch's N_1: nobody would note its velocity between the sub-steps", n);
    for (int e = 0; e < Ne; ++e) REQUIRE(written[e] == 1, "element %d is written by %d patches", e, written[e]);
    out[0] = nP; out[1] = plan.nG; out[2] = plan.nBand; out[3] = plan.P; out[4] = (int64_t)plan.lds; out[5] = sumE2; out[6] = sumN2; out[7] = orphan_patches;
    return 0;
} catch (...) { return caught(msg, msg_cap); }

// The Hilbert order alone: any coordinates (NaN, infinities, all equal) must give a permutation
// nxs_dyn_set_halo's padding of one-directional neighbours (nxs_cut::pad_halo_directions): lists in, padded lists out (capacity cap entries each); returns 0,
// 1 (msg: what is wrong with the lists) or a guard code.  *ns / *nr are updated.
int pc_pad_halo(int32_t *send_procs, int32_t *send_offsets, int *ns, int32_t *recv_procs, int32_t *recv_offsets, int *nr, int cap, char *msg, int msg_cap) try {
    put_msg(msg, msg_cap, "");
    std::vector<int> sp(send_procs, send_procs + *ns), so(send_offsets, send_offsets + *ns + 1), rp(recv_procs, recv_procs + *nr), ro(recv_offsets, recv_offsets + *nr + 1);
    const std::vector<int> sp0(sp), so0(so), rp0(rp), ro0(ro);
    const std::string bad = pad_halo_directions(sp, so, rp, ro);
    if (!bad.empty()) { put_msg(msg, msg_cap, bad); return 1; }
    // what set_halo relies on: the caller's entries untouched in front, every partner in both lists, the added segments empty, totals unchanged
    REQUIRE(std::equal(sp0.begin(), sp0.end(), sp.begin()) && std::equal(so0.begin(), so0.end(), so.begin()), "the caller's send entries moved%s", "");
    REQUIRE(std::equal(rp0.begin(), rp0.end(), rp.begin()) && std::equal(ro0.begin(), ro0.end(), ro.begin()), "the caller's receive entries moved%s", "");
    { std::vector<int> a(sp), b(rp); std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end()); REQUIRE(a == b, "the two lists still differ%s", ""); }
    for (size_t k = sp0.size(); k < sp.size(); ++k) REQUIRE(so[k + 1] == so[k] && so[k] == so0.back(), "added send segment %zu is not empty", k);
    for (size_t k = rp0.size(); k < rp.size(); ++k) REQUIRE(ro[k + 1] == ro[k] && ro[k] == ro0.back(), "added receive segment %zu is not empty", k);
    REQUIRE((int)sp.size() <= cap && (int)rp.size() <= cap, "capacity %d too small", cap);
    std::copy(sp.begin(), sp.end(), send_procs); std::copy(so.begin(), so.end(), send_offsets); *ns = (int)sp.size();
    std::copy(rp.begin(), rp.end(), recv_procs); std::copy(ro.begin(), ro.end(), recv_offsets); *nr = (int)rp.size();
    return 0;
} catch (...) { return caught(msg, msg_cap); }

int pc_hilbert(const double *x, const double *y, int n, int32_t *order_out, char *msg, int msg_cap) try {
    std::vector<int> order;
    hilbert_order(x, y, n, order);
    REQUIRE((int)order.size() == std::max(n, 0), "size");
    std::vector<char> seen((size_t)std::max(n, 0), 0);
    for (int v : order) { REQUIRE(v >= 0 && v < n && !seen[v], "not a permutation at %d", v); seen[v] = 1; }
    for (int i = 0; i < n; ++i) order_out[i] = order[i];
    return 0;
} catch (...) { return caught(msg, msg_cap); }

// bamg's convex completion (nxs_hull.inl).  out: [0] ok, [1] fill triangles, [2] hull edges.  When it is built: the fill triangles are
// counter-clockwise and, with the mesh, cover the hull polygon exactly (integer areas).
int pc_hull(const int32_t *index, const double *x, const double *y, int nods, int nels, int mode, int64_t *out, char *msg, int msg_cap) try {
    put_msg(msg, msg_cap, "");
    out[0] = out[1] = out[2] = 0;
    std::vector<int> ix, iy;
    double coef, px, py;
    if (!nxs_hull::int_plane(x, y, nods, ix, iy, coef, px, py)) { put_msg(msg, msg_cap, "coefIcoor should be positive"); return 2; }
    const nxs_hull::Completion c = nxs_hull::complete_any(index, ix.data(), iy.data(), nods, nels, mode);
    out[0] = c.ok;
    if (!c.ok) { put_msg(msg, msg_cap, c.why); return 0; }
    out[1] = (int64_t)c.fill.size() / 3; out[2] = (int64_t)c.hull.size();
    const nxs_hull::Pts P{ix.data(), iy.data()};
    typedef __int128 i128;
    i128 area_mesh = 0, area_fill = 0, area_hull = 0;
    for (int e = 0; e < nels; ++e) area_mesh += P.orient(index[3 * e] - 1, index[3 * e + 1] - 1, index[3 * e + 2] - 1);
    {   // every hull edge names the triangle behind it and the local edge (vertices VOTE[k]) that IS the hull edge
        static const int VOTE[3][2] = {{1, 2}, {2, 0}, {0, 1}};
        for (const auto &h : c.hull) {
            int tv[3];
            if (h.tri < nels) { for (int k = 0; k < 3; ++k) tv[k] = index[3 * h.tri + k] - 1; }
            else { REQUIRE(h.tri - nels < (int)c.fill.size() / 3, "hull edge names fill triangle %d", h.tri - nels); for (int k = 0; k < 3; ++k) tv[k] = c.fill[3 * (size_t)(h.tri - nels) + k]; }
            REQUIRE(h.k >= 0 && h.k < 3 && tv[VOTE[h.k][0]] == h.a && tv[VOTE[h.k][1]] == h.b, "hull edge %d -> %d is not edge %d of triangle %d", h.a, h.b, h.k, h.tri);
        }
    }
    for (size_t i = 0; i + 2 < c.fill.size(); i += 3) {
        const long long a = P.orient(c.fill[i], c.fill[i + 1], c.fill[i + 2]);
        REQUIRE(a > 0, "fill triangle %zu is not counter-clockwise", i / 3);
        area_fill += a;
    }
    for (const auto &h : c.hull) {
        area_hull += (i128)ix[h.a] * iy[h.b] - (i128)ix[h.b] * iy[h.a];
        REQUIRE(h.tri >= 0 && h.tri < nels + (int)c.fill.size() / 3 && h.k >= 0 && h.k < 3, "hull edge %d -> %d: triangle %d", h.a, h.b, h.tri);
    }
    REQUIRE(area_mesh + area_fill == area_hull, "mesh + fill triangles do not cover the hull (%lld + %lld vs %lld)", (long long)area_mesh, (long long)area_fill, (long long)area_hull);
    return 0;
} catch (...) { return caught(msg, msg_cap); }

// the guard itself: a std::bad_alloc, a std::length_error and a foreign exception become status codes
int pc_guard_selftest(int which, char *msg, int msg_cap) try {
    if (which == 0) { std::vector<double> v; v.resize(v.max_size() + 1); }
    if (which == 1) throw std::bad_alloc();
    if (which == 2) throw std::runtime_error("boom");
    if (which == 3) throw 42;
    return 0;
} catch (...) { return nxs_guard::caught("pc_guard_selftest", [&](int, const char *t) { put_msg(msg, msg_cap, t); }); }

}  // extern "C"
