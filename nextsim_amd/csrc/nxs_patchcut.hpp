// nxs_patchcut.hpp -- HOST-ONLY part of libnxsdyn.so's mesh preparation: cutting a rank's mesh into the node patches of the fused
// sub-step kernels (one ring of halo: HostPatches; D rings: HostPatches2), the node-ring patches of the open-water smoother, the
// tables of the halo exchange inside the kernels and of the resident sub-step loop.  Plain C++17, no HIP: nxs_dyn.hip includes it for
// the product, tests/native/patchcut_host.cpp compiles the very same text with -fsanitize=address,undefined (tests/test_sanitizers.py)
// -- nxs_dyn_set_mesh runs this code after every regrid (FE.cpp:3071-3154 -> distributedMeshProcessing, FE.cpp:50-143), on whatever
// numbering and partition the host hands over, so it is the host code of the library that sees the most varied input.
// Every function either fills its output completely or reports why not (a std::string); none of them touches the device.
#ifndef NXS_PATCHCUT_HPP
#define NXS_PATCHCUT_HPP

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

// limits shared with the kernels (nxs_dyn_kernels.inl defines the same values; nxs_dyn.hip static_asserts the agreement)
#define NXS_CUT_BLOCK 256         // BLOCK: nodes per block of the per-node kernels (k_smooth_halo's send blocks, the open-water block flags)
#define NXS_CUT_T256_MAXP 128     // NXS_T256_MAXP: patches of up to this many own nodes run in 256-thread workgroups
#define NXS_CUT_RES_NBR 24        // NXS_RES_NBR: neighbour patches a resident patch can wait for
#define NXS_CUT_RES_MAXNB 16      // NXS_RES_MAXNB: neighbour ranks of the several-rank resident kernel
#define NXS_CUT_RES_EPT 1         // elements per thread k_substep_resident holds (512 threads: patches of up to 512 elements, two workgroups per CU)
#define NXS_CUT_RESB_EPT 4        // NXS_RESB_EPT / NXS_RESB_NPT: elements / own nodes per thread of k_substep_resident_big (one workgroup per CU:
#define NXS_CUT_RESB_NPT 2        //   patches of up to 2048 elements and 1024 own nodes, corner slots in ten bits: at most 1024 staged nodes)

namespace nxs_cut {

struct HostPatches {
    int nP = 0, Pmax = 0, Emax = 0, Mmax = 0, Wp = 0;
    std::vector<int> own_cnt, elem_cnt, node_cnt, pnodes, pelem;
    std::vector<unsigned short> ptri, pfan;
    double avg_elems_per_own_node = 0.;
    bool used_hilbert = false;  // the caller's numbering had no locality: patches cut along a Hilbert curve
};

struct HostPatches2 {
    int nP = 0, D = 0, NDmax = 0, NSmax = 0, EDmax = 0, ESmax = 0, Wp = 0;
    std::vector<int> ncnt, ecnt, pnodes, pelem;
    std::vector<unsigned short> ptri, pfan;
};

// The mesh as the cutters see it: 0-based corner arrays, the ghost flag of every corner, undisplaced coordinates.
struct MeshView {
    const std::vector<int> *t;      // t[0], t[1], t[2]: [Ne]
    const unsigned char *ghost3;    // [3 Ne]
    const double *x0, *y0;          // [Nn]
    int Nn, Ne, No;
    const char *sent = nullptr;     // [No] several ranks, once the halo lists are known: != 0 = an own node this rank sends to a neighbour (else NULL)
};

// initUpdateGhosts' neighbour lists made mutual (nxs_dyn_set_halo; FE.cpp:14003-14088 leaves M_recipients_proc_id and M_local_ghosts_proc_id as two sets that a
// ragged partition makes differ): every rank of one list that is missing from the other is APPENDED there with an empty segment -- behind the caller's neighbours,
// whose numbers and offsets stay what they were.  Both ranks of a one-directional link do this from their own lists and so agree without talking.  Returns "" or what
// is wrong with the lists (a rank named twice; offsets that do not match).  Host only; tests/native/patchcut_host.cpp::pc_pad_halo.
inline std::string pad_halo_directions(std::vector<int> &send_procs, std::vector<int> &send_offsets, std::vector<int> &recv_procs, std::vector<int> &recv_offsets) {
    if (send_offsets.size() != send_procs.size() + 1 || recv_offsets.size() != recv_procs.size() + 1) return "offsets do not match the neighbour lists";
    for (int side = 0; side < 2; ++side) {   // a neighbour is named once per list (the padding and the mailbox's flag slots rely on it)
        std::vector<int> v(side ? recv_procs : send_procs);
        std::sort(v.begin(), v.end());
        const auto dup = std::adjacent_find(v.begin(), v.end());
        if (dup != v.end()) return std::string(side ? "recv" : "send") + "_procs names rank " + std::to_string(*dup) + " twice";
    }
    const std::vector<int> sp0(send_procs), rp0(recv_procs);
    const int ts = send_offsets.back(), tr = recv_offsets.back();
    for (int q : rp0) if (std::find(sp0.begin(), sp0.end(), q) == sp0.end()) { send_procs.push_back(q); send_offsets.push_back(ts); }
    for (int q : sp0) if (std::find(rp0.begin(), rp0.end(), q) == rp0.end()) { recv_procs.push_back(q); recv_offsets.push_back(tr); }
    return "";
}

// node -> elements CSR, ascending element number per node
inline void node_fans(const std::vector<int> t[3], int Nn, int Ne, std::vector<int> &off, std::vector<int> &adj) {
    off.assign((size_t)Nn + 1, 0);
    for (int k = 0; k < 3; ++k) for (int e = 0; e < Ne; ++e) off[t[k][e] + 1]++;
    for (int n = 0; n < Nn; ++n) off[n + 1] += off[n];
    adj.resize((size_t)off[Nn]);
    std::vector<int> fill(off.begin(), off.end() - 1);
    for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) adj[fill[t[k][e]]++] = e;
}

// order: owned nodes in the order they are cut into patches of P.
// Ecap > 0: a patch is closed early when one more own node would take it past Ecap elements (the resident kernel holds a fixed number of
// elements per thread; a partition whose own nodes are not contiguous along the numbering -- an RCB part of a Hilbert-numbered mesh --
// otherwise has a few patches of two distant blobs with 1.5 times the elements of the others, and the whole round waits for them).
// n_first > 0: the first n_first nodes of `order` are cut into patches of P_first nodes, the others into patches of P (several ranks: the nodes
// along the partition boundary in small patches of their own -- a boundary patch pays the longer exchange between ranks every sub-step of the
// resident loop, so it gets a shorter compute phase; see plan_patches)
inline bool build_patches_from_order(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int No, int P,
                                     const std::vector<int> &order, HostPatches &out, int Ecap = 0, int Mcap = 0, int n_first = 0, int P_first = 0) {
    if (P < 1 || No < 0 || No > Nn || (int)order.size() < No) return false;
    if (n_first <= 0 || P_first < 1 || n_first > No) { n_first = 0; P_first = P; }
    std::vector<int> off, adj;
    node_fans(t, Nn, Ne, off, adj);

    std::vector<int> pstart;  // patch q owns order[pstart[q] .. pstart[q + 1])
    if (Ecap <= 0) {
        for (int a = 0; a < n_first; a += P_first) pstart.push_back(a);
        for (int a = n_first; a < No; a += P) pstart.push_back(a);
    } else {
        // Mcap > 0: ... or past Mcap staged nodes (own + halo; the large-patch resident kernel names a corner's slot in ten bits)
        std::vector<int> seen(Ne, -1), seen_n, trial_n;
        if (Mcap > 0) { seen_n.assign(Nn, -1); trial_n.assign(Nn, -1); }
        int cnt_n = 0, cnt_e = 0, cnt_m = 0, q = 0, stamp = 0;
        auto fresh_nodes = [&](int n) {  // nodes the elements of n that are new to patch q would add to its staged nodes
            if (Mcap <= 0) return 0;
            ++stamp;
            int c = 0;
            for (int j = off[n]; j < off[n + 1]; ++j) {
                const int e = adj[j];
                if (seen[e] == q) continue;
                for (int k = 0; k < 3; ++k) {
                    const int v = t[k][e];
                    if (seen_n[v] != q && trial_n[v] != stamp) { trial_n[v] = stamp; ++c; }
                }
            }
            return c;
        };
        if (No > 0) pstart.push_back(0);
        for (int i = 0; i < No; ++i) {
            const int n = order[i];
            int fresh = 0;
            for (int j = off[n]; j < off[n + 1]; ++j) fresh += seen[adj[j]] != q ? 1 : 0;
            int fresh_m = fresh_nodes(n);
            // close the patch before this node: it is full (of the size its part of the order takes), the small patches end here, or a cap would be passed
            if (cnt_n > 0 && (cnt_n >= (i < n_first ? P_first : P) || (n_first > 0 && i == n_first) || cnt_e + fresh > Ecap || (Mcap > 0 && cnt_m + fresh_m > Mcap))) {
                pstart.push_back(i);
                ++q; cnt_n = 0; cnt_e = 0; cnt_m = 0;
                fresh = off[n + 1] - off[n];
                fresh_m = fresh_nodes(n);
            }
            for (int j = off[n]; j < off[n + 1]; ++j) {
                const int e = adj[j];
                if (seen[e] == q) continue;
                seen[e] = q;
                if (Mcap > 0) for (int k = 0; k < 3; ++k) seen_n[t[k][e]] = q;
            }
            ++cnt_n; cnt_e += fresh; cnt_m += fresh_m;
        }
    }
    const int nNodePatches = (int)pstart.size();
    pstart.push_back(No);
    std::vector<int> patch_of(Nn, -1);
    for (int q = 0; q + 1 < (int)pstart.size(); ++q)
        for (int i = pstart[q]; i < pstart[q + 1]; ++i) patch_of[order[i]] = q;
    // writer patch of an element = smallest patch id among its owned nodes; none -> orphan
    std::vector<int> writer(Ne, -1);
    std::vector<int> orphans;
    for (int e = 0; e < Ne; ++e) {
        int w = -1;
        for (int k = 0; k < 3; ++k) {
            const int q = patch_of[t[k][e]];
            if (q >= 0 && (w < 0 || q < w)) w = q;
        }
        writer[e] = w;
        if (w < 0) orphans.push_back(e);
    }
    const int EORPH = 2 * P;
    const int nOrphPatches = ((int)orphans.size() + EORPH - 1) / EORPH;
    const int nP = nNodePatches + nOrphPatches;

    std::vector<std::vector<int>> pel(nP), pnd(nP);
    std::vector<int> own_cnt(nP, 0);
    std::vector<int> mark(Ne, -1), slot_of(Nn, -1);
    size_t tot_e = 0;
    for (int q = 0; q < nNodePatches; ++q) {
        const int a = pstart[q], bnd = pstart[q + 1];
        own_cnt[q] = bnd - a;
        auto &el = pel[q];
        for (int i = a; i < bnd; ++i) {
            const int n = order[i];
            for (int j = off[n]; j < off[n + 1]; ++j) {
                const int e = adj[j];
                if (mark[e] != q) { mark[e] = q; el.push_back(e); }
            }
        }
        std::sort(el.begin(), el.end());
        tot_e += el.size();
    }
    for (int q = 0; q < nOrphPatches; ++q) {
        auto &el = pel[nNodePatches + q];
        const int a = q * EORPH, bnd = std::min((int)orphans.size(), a + EORPH);
        el.assign(orphans.begin() + a, orphans.begin() + bnd);  // already ascending
    }
    int Emax = 0, Mmax = 0, Wp = 0, Pmax = 0;
    std::vector<std::vector<unsigned short>> tri_l(nP);
    std::vector<std::vector<std::vector<unsigned short>>> fan_l(nP);
    for (int q = 0; q < nP; ++q) {
        auto &nd = pnd[q];
        if (q < nNodePatches) {
            const int a = pstart[q];
            for (int i = 0; i < own_cnt[q]; ++i) { nd.push_back(order[a + i]); slot_of[order[a + i]] = i; }
        }
        std::vector<int> halo;
        for (int e : pel[q])
            for (int k = 0; k < 3; ++k) {
                const int n = t[k][e];
                if (slot_of[n] == -1) { slot_of[n] = -2; halo.push_back(n); }
            }
        std::sort(halo.begin(), halo.end());
        for (int n : halo) { slot_of[n] = (int)nd.size(); nd.push_back(n); }
        if (nd.size() > 65535 || pel[q].size() > 8191) return false;
        auto &tl = tri_l[q];
        tl.resize(4 * pel[q].size());
        auto &fl = fan_l[q];
        fl.assign(own_cnt[q], {});
        for (size_t l = 0; l < pel[q].size(); ++l) {
            const int e = pel[q][l];
            for (int k = 0; k < 3; ++k) {
                const int n = t[k][e], sl = slot_of[n];
                tl[4 * l + k] = (unsigned short)sl;
                if (sl < own_cnt[q]) fl[sl].push_back((unsigned short)((l << 3) | (ghost3[3 * (size_t)e + k] ? 4 : 0) | k));
            }
            tl[4 * l + 3] = 0;
        }
        for (auto &f : fl) Wp = std::max(Wp, (int)f.size());
        for (int n : nd) slot_of[n] = -1;
        Emax = std::max(Emax, (int)pel[q].size());
        Mmax = std::max(Mmax, (int)nd.size());
        Pmax = std::max(Pmax, own_cnt[q]);
    }
    Emax = (Emax + 1) & ~1;  // keep the ushort4 / double rows 16-byte aligned
    Mmax = (Mmax + 1) & ~1;
    Pmax = std::max(Pmax, 1);
    Wp = std::max(Wp, 1);
    out = HostPatches{};
    out.nP = nP; out.Pmax = Pmax; out.Emax = Emax; out.Mmax = Mmax; out.Wp = Wp;
    out.own_cnt = own_cnt;
    out.elem_cnt.resize(nP); out.node_cnt.resize(nP);
    out.pnodes.assign((size_t)nP * Mmax, 0);
    out.pelem.assign((size_t)nP * Emax, 0);
    out.ptri.assign((size_t)nP * Emax * 4, 0);
    out.pfan.assign((size_t)nP * Wp * Pmax, 0xFFFF);
    for (int q = 0; q < nP; ++q) {
        out.elem_cnt[q] = (int)pel[q].size();
        out.node_cnt[q] = (int)pnd[q].size();
        std::copy(pnd[q].begin(), pnd[q].end(), out.pnodes.begin() + (size_t)q * Mmax);
        for (size_t l = 0; l < pel[q].size(); ++l) {
            const int e = pel[q][l];
            const bool is_writer = (writer[e] == q) || (writer[e] < 0);  // orphans are written by their orphan patch
            out.pelem[(size_t)q * Emax + l] = is_writer ? e : ~e;
        }
        std::copy(tri_l[q].begin(), tri_l[q].end(), out.ptri.begin() + (size_t)q * Emax * 4);
        for (int i = 0; i < own_cnt[q]; ++i)
            for (size_t k = 0; k < fan_l[q][i].size(); ++k)
                out.pfan[(size_t)q * Wp * Pmax + k * Pmax + i] = fan_l[q][i][k];
    }
    out.avg_elems_per_own_node = No > 0 ? (double)tot_e / No : 0.;
    return true;
}

// the first n nodes sorted along a Hilbert curve through their coordinates (a non-finite coordinate sorts as the lower-left corner)
inline void hilbert_order(const double *x0, const double *y0, int n_nodes, std::vector<int> &order) {
    order.resize((size_t)std::max(n_nodes, 0));
    for (int i = 0; i < n_nodes; ++i) order[i] = i;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int n = 0; n < n_nodes; ++n) {
        if (!(x0[n] >= -1e300 && x0[n] <= 1e300 && y0[n] >= -1e300 && y0[n] <= 1e300)) continue;
        xmin = std::min(xmin, x0[n]); xmax = std::max(xmax, x0[n]); ymin = std::min(ymin, y0[n]); ymax = std::max(ymax, y0[n]);
    }
    const double ext = std::max(xmax - xmin, ymax - ymin);
    const double sc = (ext > 0. && ext <= 1e300) ? 65535. / ext : 0.;
    auto hilbert = [](unsigned x, unsigned y) {
        unsigned long long d = 0;
        for (unsigned s2 = 1u << 15; s2 > 0; s2 >>= 1) {
            const unsigned rx = (x & s2) ? 1u : 0u, ry = (y & s2) ? 1u : 0u;
            d += (unsigned long long)s2 * s2 * ((3u * rx) ^ ry);
            if (ry == 0) {
                if (rx == 1) { x = s2 - 1 - x; y = s2 - 1 - y; }  // (only the bits below s2 are looked at again: the wrap-around above them is harmless)
                const unsigned t2 = x; x = y; y = t2;
            }
        }
        return d;
    };
    auto cell = [&](double v, double lo) -> unsigned {  // [0, 65535]; the conversion of a NaN or of a value beyond unsigned's range would be undefined
        const double c = (v - lo) * sc;
        return (c >= 0. && c <= 65535.) ? (unsigned)c : (c > 65535. ? 65535u : 0u);
    };
    std::vector<unsigned long long> key((size_t)std::max(n_nodes, 0));
    for (int n = 0; n < n_nodes; ++n) key[n] = hilbert(cell(x0[n], xmin), cell(y0[n], ymin));
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return key[a] < key[b2]; });
}

// band_P > 0 (several ranks): the own nodes that share an element with a ghost node AND the own nodes this rank sends (sent: known once the halo lists are; the
// two sets differ -- a neighbour's OWNED element all of whose nodes are mine makes them its ghosts, and no ghost of mine is near: at 2 km / 8 ranks 123 of 914 sent
// nodes, scattered over 37 ordinary patches that then sat in the exchange's critical path at full size) lead the order and are cut into patches of band_P nodes
inline bool build_patches(const std::vector<int> t[3], const unsigned char *ghost3, const double *x0, const double *y0, int Nn, int Ne,
                          int No, int P, HostPatches &out, int Ecap = 0, int Mcap = 0, int band_P = 0, const char *sent = nullptr) {
    // 1st try: the caller's node numbering (keeps the patch's nodal accesses contiguous)
    std::vector<int> order(No);
    for (int i = 0; i < No; ++i) order[i] = i;
    std::vector<char> band;
    int n_band = 0;
    if (band_P > 0 && No < Nn) {
        band.assign(No, 0);
        for (int e = 0; e < Ne; ++e) {
            const int v[3] = {t[0][e], t[1][e], t[2][e]};
            if (v[0] < No && v[1] < No && v[2] < No) continue;
            for (int k = 0; k < 3; ++k) if (v[k] < No) band[v[k]] = 1;
        }
        if (sent) for (int i = 0; i < No; ++i) if (sent[i]) band[i] = 1;
        for (int i = 0; i < No; ++i) n_band += band[i];
        std::stable_partition(order.begin(), order.end(), [&](int n) { return band[n] != 0; });
    }
    bool ok = build_patches_from_order(t, ghost3, Nn, Ne, No, P, order, out, Ecap, Mcap, n_band, band_P);
    if (ok && out.avg_elems_per_own_node <= 3.0 + (n_band > 0 ? 0.5 : 0.)) return true;
    // numbering without locality: cut patches along a Hilbert curve through the node coordinates
    // (consecutive runs of a Hilbert curve are compact blobs: small halos)
    hilbert_order(x0, y0, No, order);
    if (n_band > 0) std::stable_partition(order.begin(), order.end(), [&](int n) { return band[n] != 0; });
    HostPatches alt;
    if (build_patches_from_order(t, ghost3, Nn, Ne, No, P, order, alt, Ecap, Mcap, n_band, band_P) && (!ok || alt.avg_elems_per_own_node < out.avg_elems_per_own_node)) {
        out = std::move(alt);
        out.used_hilbert = true;
        return true;
    }
    return ok;
}

// D-ring patches of k_substep_multi (DevPatches2); single rank (every node owned, no orphan elements).
inline bool build_patches2(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int P, int D, const std::vector<int> &order, HostPatches2 &out) {
    if (P < 1 || D < 1 || (int)order.size() < Nn) return false;
    std::vector<int> off, adj;
    node_fans(t, Nn, Ne, off, adj);
    const int nP = (Nn + P - 1) / P;
    std::vector<int> patch_of(Nn, -1);
    for (int i = 0; i < Nn; ++i) patch_of[order[i]] = i / P;
    std::vector<int> writer(Ne);
    for (int e = 0; e < Ne; ++e) writer[e] = std::min({patch_of[t[0][e]], patch_of[t[1][e]], patch_of[t[2][e]]});

    out = HostPatches2{};
    out.nP = nP; out.D = D;
    out.ncnt.assign((size_t)nP * (D + 1), 0); out.ecnt.assign((size_t)nP * D, 0);
    std::vector<std::vector<int>> pel(nP), pnd(nP);
    std::vector<std::vector<unsigned short>> tri_l(nP);
    std::vector<std::vector<std::vector<unsigned short>>> fan_l(nP);
    std::vector<int> emark(Ne, -1), eslot(Ne, -1), slot_of(Nn, -1);
    for (int q = 0; q < nP; ++q) {
        const int a = q * P, bnd = std::min(Nn, a + P);
        auto &nd = pnd[q];
        auto &el = pel[q];
        int *nc = out.ncnt.data() + (size_t)q * (D + 1), *ec = out.ecnt.data() + (size_t)q * D;
        for (int i = a; i < bnd; ++i) { slot_of[order[i]] = (int)nd.size(); nd.push_back(order[i]); }
        nc[0] = bnd - a;
        int n_prev = 0, e_prev = 0;
        for (int lev = 1; lev <= D; ++lev) {
            // E_lev: the elements touching N_(lev-1) that are not listed yet, ascending
            std::vector<int> add;
            for (int i = n_prev; i < nc[lev - 1]; ++i)
                for (int j = off[nd[i]]; j < off[nd[i] + 1]; ++j) {
                    const int e = adj[j];
                    if (emark[e] != q) { emark[e] = q; add.push_back(e); }
                }
            std::sort(add.begin(), add.end());
            el.insert(el.end(), add.begin(), add.end());
            ec[lev - 1] = (int)el.size();
            // N_lev: their nodes that are not listed yet, ascending
            std::vector<int> addn;
            for (int l = e_prev; l < ec[lev - 1]; ++l)
                for (int k = 0; k < 3; ++k) {
                    const int n = t[k][el[l]];
                    if (slot_of[n] == -1) { slot_of[n] = -2; addn.push_back(n); }
                }
            std::sort(addn.begin(), addn.end());
            for (int n : addn) { slot_of[n] = (int)nd.size(); nd.push_back(n); }
            nc[lev] = (int)nd.size();
            n_prev = nc[lev - 1]; e_prev = ec[lev - 1];
        }
        if (nd.size() > 65535 || el.size() > 8191) {
            for (int n : nd) slot_of[n] = -1;
            return false;
        }
        for (size_t l = 0; l < el.size(); ++l) eslot[el[l]] = (int)l;
        auto &tl = tri_l[q];
        tl.assign(4 * el.size(), 0);
        for (size_t l = 0; l < el.size(); ++l)
            for (int k = 0; k < 3; ++k) tl[4 * l + k] = (unsigned short)slot_of[t[k][el[l]]];
        const int nsolved = nc[D - 1];
        auto &fl = fan_l[q];
        fl.assign(nsolved, {});
        for (int i = 0; i < nsolved; ++i) {
            const int n = nd[i];
            for (int j = off[n]; j < off[n + 1]; ++j) {  // ascending element id = the order of the serial scatter
                const int e = adj[j];
                int k = 0;
                while (t[k][e] != n) ++k;
                fl[i].push_back((unsigned short)((eslot[e] << 3) | (ghost3[3 * (size_t)e + k] ? 4 : 0) | k));
            }
            out.Wp = std::max(out.Wp, (int)fl[i].size());
        }
        for (int n : nd) slot_of[n] = -1;
        out.NDmax = std::max(out.NDmax, nc[D]); out.NSmax = std::max(out.NSmax, nc[D - 1]);
        out.EDmax = std::max(out.EDmax, ec[D - 1]); out.ESmax = std::max(out.ESmax, D >= 2 ? ec[D - 2] : 0);
    }
    out.NDmax = (out.NDmax + 1) & ~1; out.NSmax = (out.NSmax + 1) & ~1; out.EDmax = (out.EDmax + 1) & ~1; out.ESmax = std::max(2, (out.ESmax + 1) & ~1);
    out.Wp = std::max(out.Wp, 1);
    out.pnodes.assign((size_t)nP * out.NDmax, 0);
    out.pelem.assign((size_t)nP * out.EDmax, 0);
    out.ptri.assign((size_t)nP * out.EDmax * 4, 0);
    out.pfan.assign((size_t)nP * out.Wp * out.NSmax, 0xFFFF);
    for (int q = 0; q < nP; ++q) {
        std::copy(pnd[q].begin(), pnd[q].end(), out.pnodes.begin() + (size_t)q * out.NDmax);
        for (size_t l = 0; l < pel[q].size(); ++l) {
            const int e = pel[q][l];
            out.pelem[(size_t)q * out.EDmax + l] = (writer[e] == q) ? e : ~e;
        }
        std::copy(tri_l[q].begin(), tri_l[q].end(), out.ptri.begin() + (size_t)q * out.EDmax * 4);
        for (size_t i = 0; i < fan_l[q].size(); ++i)
            for (size_t k = 0; k < fan_l[q][i].size(); ++k)
                out.pfan[(size_t)q * out.Wp * out.NSmax + k * out.NSmax + i] = fan_l[q][i][k];
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Which patch size: the single-ring patches of k_substep_fused / k_substep_resident (what upload_patches uploads).
struct PatchPlan {
    HostPatches hp;
    int P = 0;
    size_t fused_lds = 0;  // staged nodes, corner forces + their pair of zeros
    bool cut_big = false;  // cut for k_substep_resident_big (one large patch per CU): not what the one-launch-per-sub-step kernel wants
};

inline size_t fused_lds_of(const HostPatches &hp) { return (4 * (size_t)hp.Mmax + 6 * (size_t)hp.Emax + 2) * sizeof(double); }
// LDS of k_substep_resident (velocities, corner forces + zeros, shape coefficients, nodal inputs, M_UM / M_UT, eight fan entries per own node;
// several ranks: + the halo slots' sources and the neighbour ranks' mailbox addresses) and of k_substep_resident_big (velocities, corner forces,
// fan entries; several ranks: + the halo slots' sources)
inline size_t resident_lds_of(const HostPatches &hp, bool multi_rank) {
    return (2 * (size_t)hp.Mmax + 12 * (size_t)hp.Emax + 14 * (size_t)hp.Pmax + 2) * sizeof(double) + 16 * (size_t)hp.Pmax +
           (multi_rank ? 16 * (size_t)std::min(hp.Mmax, 512) + 20 * (size_t)NXS_CUT_RES_MAXNB : 0);
}
inline size_t resident_big_lds_of(const HostPatches &hp, bool multi_rank) {
    return (2 * (size_t)hp.Mmax + 6 * (size_t)hp.Emax + 2) * sizeof(double) + 16 * (size_t)hp.Pmax + (multi_rank ? 16 * (size_t)hp.Mmax + 20 * (size_t)NXS_CUT_RES_MAXNB : 0);
}
inline bool resident_is_big(const HostPatches &hp) { return hp.Emax > 512 * NXS_CUT_RES_EPT || hp.Pmax > 512; }

// patch_nodes > 0: the caller's size (shrunk until it fits); else automatic.  want_resident: option fused = 4 was set before set_mesh (the mesh is
// then cut for ONE round of resident 512-thread workgroups where that is possible), res_ept: elements per thread the resident kernel holds.
// cus: compute units of the device.  Returns "" or the reason it failed.
inline std::string plan_patches(const MeshView &m, int patch_nodes, bool want_resident, int cus, PatchPlan &out, int res_ept = NXS_CUT_RES_EPT, bool allow_big = true,
                                int band_nodes = 0 /* several ranks, resident loop: the nodes along the partition boundary in patches of this size (0: like the others) */) {
    char msg[160];
    HostPatches &hp = out.hp;
    int P = 0;
    out.fused_lds = 0;
    auto build = [&](int PP, int Ecap_big = 0) -> bool {
        // patches of up to ~200 nodes hold one element per thread of a 512-thread workgroup (k_substep_resident requires it, and one round
        // of the one-launch-per-sub-step kernel is as slow as its largest patch): none may exceed 480 elements
        const int Ecap = Ecap_big > 0 ? Ecap_big : (PP > NXS_CUT_T256_MAXP && PP <= 208) ? 480 : 0;
        if (!build_patches(m.t, m.ghost3, m.x0, m.y0, m.Nn, m.Ne, m.No, PP, hp, Ecap, Ecap_big > 0 ? 1000 : 0, (want_resident && Ecap_big == 0 && band_nodes < PP) ? band_nodes : 0, m.sent)) return false;
        out.fused_lds = fused_lds_of(hp);
        return true;
    };
    out.cut_big = false;
    cus = std::max(cus, 1);
    if (patch_nodes > 0) {
        P = std::max(64, std::min(patch_nodes, 1024));
        for (;;) {
            if (!build(P)) { snprintf(msg, sizeof msg, "patch construction failed (patch_nodes=%d)", P); return msg; }
            if ((out.fused_lds <= 80 * 1024 && hp.Mmax <= 1024) || P <= 64) break;
            P = std::max(64, P * 3 / 4);
        }
    } else {
        // Large patches recompute few halo elements; the limits are the LDS of two resident workgroups per CU
        // (160 KiB / 2) and, above all, WHOLE ROUNDS: the grid runs in rounds of `slots` resident workgroups and a
        // last round that is partly empty costs as much as a full one.  So: the smallest number of rounds k whose
        // patch size ceil(No / (k*slots)) fits, e.g. 730 k nodes -> 3 rounds of 512 patches of 476 nodes (not 2.79
        // rounds of 512-node patches); 92 k nodes (one rank of eight) -> one round of 511 patches of 180 nodes.
        const int slots512 = 2 * cus, slots256 = 4 * cus;  // 16 waves per CU (112 VGPRs): 2 x 512 or 4 x 256 threads
        bool done = false;
        if (want_resident && m.No > 0) {
            // the resident sub-step loop was asked for (before set_mesh): ONE round of 512-thread workgroups with one element per thread --
            // two patches per CU, or one of twice the size where those would be smaller than ~100 nodes (10 km, 30 k nodes: 255 patches of
            // 116 nodes 0.625 ms/step, 462 of 64 nodes 0.653) -- also where the one-launch-per-sub-step kernel would take smaller patches
            // (65 k - 90 k nodes: 256-thread workgroups, four per CU); above ~200 nodes per patch two elements per thread (a rank of four
            // of the 2 km mesh: 512 patches of ~360 nodes)
            int Pr = (int)(((long long)m.No + 2 * cus - 1) / (2 * cus));
            if (Pr < 100) Pr = (int)(((long long)m.No + cus - 1) / cus);
            Pr = std::max(64, (Pr + 3) & ~3);  // (tiny partitions: 64-node patches as the one-launch-per-sub-step kernel takes -- fewer, fuller workgroups)
            for (int it = 0; it < 4 && Pr <= 208 && !done; ++it, Pr += 4) {  // orphan patches (multi-rank) may add a few workgroups
                if (!build(Pr)) break;
                done = hp.Emax <= 512 * res_ept && hp.nP <= 2 * cus && out.fused_lds <= 80 * 1024;
                if (done) P = Pr;
            }
            // too large for that: ONE workgroup per CU with four elements and two own nodes per thread (k_substep_resident_big: the partition a
            // rank of four of the 2 km mesh holds -- 256 patches of ~720 nodes / ~1 600 elements)
            if (!done && allow_big) {
                int Pb = (int)(((long long)m.No + cus - 1) / cus);
                Pb = std::max(256, (Pb + 3) & ~3);
                for (int it = 0; it < 8 && Pb <= 512 * NXS_CUT_RESB_NPT && !done; ++it, Pb += 8) {  // orphan patches and patches closed early add a few workgroups
                    if (!build(Pb, 512 * NXS_CUT_RESB_EPT - 64)) break;
                    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] large-patch cut: P=%d -> nP=%d Pmax=%d Emax=%d Mmax=%d lds=%zu\n", Pb, hp.nP, hp.Pmax, hp.Emax, hp.Mmax, resident_big_lds_of(hp, m.No < m.Nn));
                    done = hp.Emax <= 512 * NXS_CUT_RESB_EPT && hp.Pmax <= 512 * NXS_CUT_RESB_NPT && hp.Mmax <= 1024 && hp.nP <= cus &&
                           resident_big_lds_of(hp, m.No < m.Nn) <= 160 * 1024 && out.fused_lds <= 160 * 1024;
                    if (done) { P = Pb; out.cut_big = true; }
                }
            }
        }
        for (int k = 1; k <= 64 && !done; ++k) {
            P = (int)(((long long)m.No + (long long)k * slots512 - 1) / ((long long)k * slots512));
            P = (P + 3) & ~3;
            if (P > 512) continue;
            if (P <= NXS_CUT_T256_MAXP) break;  // small mesh: the 256-thread kernel below
            for (int it = 0; it < 4 && !done; ++it) {  // orphan patches (multi-rank) may add a few workgroups
                if (it > 0) P += 4;
                if (!build(P)) { snprintf(msg, sizeof msg, "patch construction failed (patch_nodes=%d)", P); return msg; }
                if (out.fused_lds > 80 * 1024 || hp.Mmax > 1024) break;  // does not fit twice (or its corner slots do not fit 10 bits): more rounds of smaller patches
                done = hp.nP <= k * slots512;
            }
        }
        if (!done) {
            P = (int)(((long long)m.No + slots256 - 1) / slots256);
            P = std::max(64, std::min((P + 3) & ~3, NXS_CUT_T256_MAXP));
            if (!build(P)) { snprintf(msg, sizeof msg, "patch construction failed (patch_nodes=%d)", P); return msg; }
        }
    }
    out.P = P;
    if (out.fused_lds > 160 * 1024) { snprintf(msg, sizeof msg, "patches need %zu B of LDS", out.fused_lds); return msg; }
    if (hp.Mmax > 1024) { snprintf(msg, sizeof msg, "a patch stages %d nodes (at most 1024: choose smaller patches)", hp.Mmax); return msg; }
    return "";
}

// {pelem, the three corner slots in 10 bits each}: the 8-byte word per patch element that k_substep_fused streams (needs Mmax <= 1024)
inline void pack_pet(const HostPatches &hp, std::vector<int> &pet_xy) {
    pet_xy.resize(2 * (size_t)hp.nP * hp.Emax);
    for (size_t i = 0; i < (size_t)hp.nP * hp.Emax; ++i) {
        pet_xy[2 * i] = hp.pelem[i];
        pet_xy[2 * i + 1] = (int)hp.ptri[4 * i] | ((int)hp.ptri[4 * i + 1] << 10) | ((int)hp.ptri[4 * i + 2] << 20);
    }
}

// The bamg-order rows of k_prep_fused: NodalElementConnectivity of every own node of every patch with the elements named by their patch slots
// ([nP][W1][Pmax], 0xFFFF = no element: NaN pad, Q2).  n2e: [W1][Nn] global element numbers, -1 = pad.  Every element of an own node's row is in
// the patch (a patch holds all elements that touch its own nodes); a row that names an element outside it -- a caller-supplied table that does not
// belong to this mesh -- makes the function return false, and the caller keeps the two separate kernels.
inline bool build_prep_rows(const HostPatches &hp, const int *n2e, int W1, int Nn, std::vector<unsigned short> &rows) {
    rows.assign((size_t)hp.nP * W1 * hp.Pmax, 0xFFFF);
    std::vector<std::pair<int, int>> byid;
    for (int q = 0; q < hp.nP; ++q) {
        const int nE = hp.elem_cnt[q];
        byid.resize(nE);
        for (int l = 0; l < nE; ++l) { const int raw = hp.pelem[(size_t)q * hp.Emax + l]; byid[l] = {raw >= 0 ? raw : ~raw, l}; }
        std::sort(byid.begin(), byid.end());
        for (int i = 0; i < hp.own_cnt[q]; ++i) {
            const int n = hp.pnodes[(size_t)q * hp.Mmax + i];
            if (n < 0 || n >= Nn) return false;
            for (int j = 0; j < W1; ++j) {
                const int e = n2e[(size_t)j * Nn + n];
                if (e < 0) continue;
                auto it = std::lower_bound(byid.begin(), byid.end(), std::make_pair(e, -1));
                if (it == byid.end() || it->first != e) return false;
                rows[((size_t)q * W1 + j) * hp.Pmax + i] = (unsigned short)it->second;
            }
        }
    }
    return true;
}
// LDS of k_prep_fused: displaced coordinates and ssh of the staged nodes; per patch element (mass x area, C_bu), (area, drag x area), the Jacobian, the corner slots
inline size_t prep_fused_lds_of(const HostPatches &hp) {
    return 3 * (size_t)((hp.Mmax + 1) & ~1) * sizeof(double) + (size_t)hp.Emax * (16 + 16 + 8 + 8);
}

// ------------------------------------------------------------------------------------------------
// The D-ring patches of k_substep_multi and the NodalConnectivity rows of their solved nodes in patch-local slots (k_smooth_multi).
struct Patch2Plan {
    HostPatches2 hp;
    bool pair_kernel = false;  // cut for k_substep_pair (depth 2, two workgroups per CU)
    int P_fit = 0;             // k_substep_pair: the largest patch size that fits (the hint for the next mesh of the handle; P may be smaller: whole rounds)
    int P = 0, threads = 512;
    size_t lds = 0, smooth_lds = 0;
    std::vector<unsigned short> pnbr;  // [nP][W2][NSmax], empty when a caller-supplied row reaches beyond its patch
};

inline size_t multi_lds_of(const HostPatches2 &x) { return (4 * (size_t)x.NDmax + 6 * (size_t)x.EDmax + 2 + 4 * (size_t)x.ESmax) * sizeof(double); }
// k_substep_pair (two sub-steps per launch, the stresses between them in registers): staged nodes and corner forces only; its 512-thread block
// takes the elements of the first sub-step in three rounds, those of the second and the nodes of the first in two, the own nodes in one
// (+ 4: the pair of zeros pad fan entries name, and 16 bytes of control words for k_substep_flow)
inline size_t pair_lds_of(const HostPatches2 &x) { return (4 * (size_t)x.NDmax + 6 * (size_t)x.EDmax + 4) * sizeof(double); }
inline bool pair_kernel_fits(const HostPatches2 &x, int own_max, int T = 512) { return x.D == 2 && x.EDmax <= 3 * T && x.ESmax <= 2 * T && x.NSmax <= 2 * T && own_max <= T && x.NDmax <= 1024 /*corner slots travel in ten bits*/; }

// k_substep_pair's fan gather reads eight 16-byte corner forces per solved node from LDS: [3][EDmax] pairs, the pair of zeros behind them.  The first eight fan
// entries of every solved node as the LDS indices themselves (corner * EDmax + element slot; 3 * EDmax for a pad entry or a ghost corner, FE.cpp:10456), two per
// 32-bit word: out[(q * 4 + k) * NSmax + i] = index of entry 2k | index of entry 2k + 1 << 16.  (3 * EDmax + 1 <= 65 536: EDmax <= 3 * 512.)
inline void decode_fan8(const HostPatches2 &hp, std::vector<unsigned int> &out) {
    out.assign((size_t)hp.nP * 4 * hp.NSmax, 0u);
    const unsigned zidx = 3u * (unsigned)hp.EDmax;
    for (int q = 0; q < hp.nP; ++q)
        for (int i = 0; i < hp.NSmax; ++i)
            for (int k = 0; k < 4; ++k) {
                unsigned idx[2];
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int row = 2 * k + h2;
                    const unsigned ent = row < hp.Wp ? hp.pfan[((size_t)q * hp.Wp + row) * hp.NSmax + i] : 0xFFFFu;
                    idx[h2] = (ent == 0xFFFFu || (ent & 4u)) ? zidx : (ent & 3u) * (unsigned)hp.EDmax + (ent >> 3);
                }
                out[((size_t)q * 4 + k) * hp.NSmax + i] = idx[0] | (idx[1] << 16);
            }
}

// k_substep_flow (one launch per step, tasks = (pair of sub-steps, patch)): which patches must have finished pair-step k - 1 before patch q may start pair-step k.
//   q READS the state of the elements of its outer level from the buffer their WRITERS filled in the pair-step before, and the velocities of the nodes it stages
//   from the slot their OWNERS filled;
//   q OVERWRITES, for the elements it writes, the buffer that the patches holding those elements in their outer level read in the pair-step before (the state
//   ping-pongs between two buffers): the same relation the other way round;
//   q itself (another workgroup may have run its pair-step before).
// CSR over the patches, ascending.  Returns false when a node has no owner or an element no writer (not a single-rank cut).
inline bool build_flow_deps(const HostPatches2 &hp, int Nn, int Ne, std::vector<int> &ptr, std::vector<int> &dep) {
    const int nP = hp.nP, D = hp.D;
    ptr.assign((size_t)nP + 1, 0);
    dep.clear();
    if (nP <= 0 || D < 1) return false;
    std::vector<int> owner((size_t)Nn, -1), writer((size_t)Ne, -1);
    for (int q = 0; q < nP; ++q) {
        for (int i = 0; i < hp.ncnt[(size_t)q * (D + 1)]; ++i) {
            const int n = hp.pnodes[(size_t)q * hp.NDmax + i];
            if (n < 0 || n >= Nn || owner[n] >= 0) return false;
            owner[n] = q;
        }
        for (int l = 0; l < hp.ecnt[(size_t)q * D + (D - 1)]; ++l) {
            const int e = hp.pelem[(size_t)q * hp.EDmax + l];
            if (e < 0) continue;
            if (e >= Ne || writer[e] >= 0) return false;
            writer[e] = q;
        }
    }
    std::vector<std::vector<int>> R((size_t)nP);
    for (int q = 0; q < nP; ++q) {
        std::vector<int> &r = R[q];
        r.push_back(q);
        for (int i = 0; i < hp.ncnt[(size_t)q * (D + 1) + D]; ++i) {
            const int n = hp.pnodes[(size_t)q * hp.NDmax + i];
            if (n < 0 || n >= Nn || owner[n] < 0) return false;
            r.push_back(owner[n]);
        }
        for (int l = 0; l < hp.ecnt[(size_t)q * D + (D - 1)]; ++l) {
            int e = hp.pelem[(size_t)q * hp.EDmax + l];
            if (e < 0) e = ~e;
            if (e >= Ne || writer[e] < 0) return false;
            r.push_back(writer[e]);
        }
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
    }
    std::vector<std::vector<int>> back((size_t)nP);
    for (int q = 0; q < nP; ++q) for (int r : R[q]) if (r != q) back[r].push_back(q);
    for (int q = 0; q < nP; ++q) {
        std::vector<int> &r = R[q];
        r.insert(r.end(), back[q].begin(), back[q].end());
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());
        ptr[q + 1] = ptr[q] + (int)r.size();
        dep.insert(dep.end(), r.begin(), r.end());
    }
    return true;
}
// the queues of k_substep_flow: queue x serves the patches [qstart[x], qstart[x + 1]) -- the split k_substep_pair's XCD remap makes (consecutive patches are
// neighbours in space; the workgroups blockIdx = x mod 8 are dealt to one XCD)
inline void flow_queues(int nP, int qstart[9]) {
    const int q = nP >> 3, r = nP & 7;
    qstart[0] = 0;
    for (int x = 0; x < 8; ++x) qstart[x + 1] = qstart[x] + (x < r ? q + 1 : q);
}

// n2n: [W2][Nn] neighbour rows (bamg order), n2n_cnt: [Nn]
inline std::string plan_patches2(const MeshView &m, bool used_hilbert, int pair_nodes, int D, bool single_round_only, int cus,
                                 const std::vector<int> &n2n, const std::vector<int> &n2n_cnt, int W2, Patch2Plan &out, bool for_pair_kernel = false,
                                 int pair_hint = 0 /* the size kept for the previous mesh of this handle: after a regrid it usually still fits */,
                                 int pair_T = 512 /* threads of a k_substep_pair workgroup: 512 (two per CU) or 256 (four per CU) */) {
    char msg[160];
    out.pair_kernel = false;
    if (m.No != m.Nn) return "multi-sub-step patches need a single-rank mesh";
    if (D < 2 || D > 8) return "multi-sub-step patches: depth out of range";
    std::vector<int> order(m.Nn);
    for (int i = 0; i < m.Nn; ++i) order[i] = i;
    // the caller's numbering if it has locality, else the Hilbert curve the single-ring patches were cut along
    if (used_hilbert) hilbert_order(m.x0, m.y0, m.Nn, order);
    HostPatches2 &hp = out.hp;
    int P = 0;
    cus = std::max(cus, 1);
    auto own_max = [&]() { int v = 0; for (int q = 0; q < hp.nP; ++q) v = std::max(v, hp.ncnt[(size_t)q * (D + 1)]); return v; };
    if (for_pair_kernel) {
        // k_substep_pair on a mesh that streams from HBM: the LARGEST patches two workgroups per CU have the LDS for (80 KB each) -- the rings
        // of a large patch are relatively thin, and what a launch saves in traffic is paid for in ring arithmetic (2 km: 5.72 ms of sub-steps
        // at 408 nodes, 5.96 at 400 where one more wave of patches is started, 6.07 at 300; 6.7 at 420, where only one workgroup fits a CU)
        if (D != 2) return "k_substep_pair runs two sub-steps per launch";
        const size_t cap = (size_t)80 * 1024 * pair_T / 512;
        auto fits = [&](int PP) {
            if (!build_patches2(m.t, m.ghost3, m.Nn, m.Ne, PP, D, order, hp)) return false;
            return pair_lds_of(hp) <= cap && pair_kernel_fits(hp, own_max(), pair_T);
        };
        if (pair_nodes > 0) {
            P = pair_nodes;
            if (!fits(P)) { snprintf(msg, sizeof msg, "patches of %d nodes do not fit k_substep_pair (80 KB of LDS, three rounds of elements)", P); return msg; }
        } else {
            // (a mesh too small to give every CU two such patches takes smaller ones: one full round of 2 x cus workgroups)
            const int wg_per_cu = 2 * 512 / pair_T;
            const int hi0 = std::min(pair_T, std::max(68, (int)((((long long)m.Nn + wg_per_cu * cus - 1) / (wg_per_cu * cus) + 3) & ~3ll) + 4));
            int lo = 64, hi = hi0;
            // every trial cuts the whole mesh (135 ms at 1.5 M triangles): the size the previous mesh of this handle took is tried first and kept if it
            // still fits (one cut per regrid instead of nine), else the search goes on below it
            if (pair_hint >= 64 && pair_hint < hi0 && fits(pair_hint)) { P = pair_hint; lo = hi = 0; }
            else if (pair_hint >= 68 && pair_hint < hi0) hi = pair_hint;
            if (hi > 0) {
                if (!fits(lo)) return "no patch size fits k_substep_pair (node numbering without locality?)";
                int last_built = lo;
                while (hi - lo > 4) {   // bisection on the patch size (the need grows with it)
                    const int mid = ((lo + hi) / 2 + 3) & ~3;
                    if (mid >= hi) break;
                    last_built = mid;
                    if (fits(mid)) lo = mid; else hi = mid;
                }
                P = lo;
                if (last_built != P && !fits(P)) return "no patch size fits k_substep_pair";
            }
            out.P_fit = P;
            // One round of workgroups or less: a whole round (a round that is a fraction full costs almost a full one).
            const int slots = wg_per_cu * cus, k = (hp.nP + slots - 1) / slots;
            if (k == 1 && hp.nP > 0 && hp.nP != k * slots) {
                int Pr = (int)((((long long)m.Nn + (long long)k * slots - 1) / ((long long)k * slots) + 3) & ~3ll);
                Pr = std::max(64, Pr);
                bool tried = false;
                for (int it = 0; it < 6 && Pr < P; ++it, Pr += 4) {   // (patches closed early add a few: a size or two up until they fit the rounds)
                    tried = true;
                    if (fits(Pr) && hp.nP <= k * slots) { P = Pr; break; }
                }
                if (tried && P == out.P_fit && !fits(P)) return "no patch size fits k_substep_pair";   // (the trials rebuilt hp: back to the size kept)
            } else if (k >= 2) {
                // More than one round: the time of a launch steps up whenever the number of patches crosses a multiple of HALF a round (one workgroup per CU: a CU then
                // runs one more patch behind the others), and falls slowly towards the next multiple (smaller patches, as many per CU).  2 km, sub-steps per step: 1 707
                // patches of 428 nodes 4.28 ms, 1 756-1 773 of 416-412 4.22, 1 790 of 408 4.25 | 1 808 of 404 4.51 ... 2 029 of 360 4.40 | 2 123 of 344 4.62
                // (profiles/r05_experiments/r5_pair_nodes_2km.log).  So: the smallest patches that stay under the multiple the largest ones are under, 1.5 % short of it.
                // Two rounds likewise: 493 k triangles, ms per step: 1 015 patches of 244 nodes (two whole rounds) 1.88, 774 of 320 2.05 | 755 of 328 (three half
                // rounds) 1.82, 737 of 336 1.91 (r5_single_492k.log); 730 k: 1 018 of 360 2.43 ms of sub-steps, 944 of 388 2.47 (four half rounds = two whole ones).
                const int unit = std::max(slots / 2, 1), hr = (hp.nP + unit - 1) / unit;
                const long long target = (long long)(0.985 * (double)hr * unit);
                int Pr = (int)((((long long)m.Nn + target - 1) / std::max(target, 1ll) + 3) & ~3ll);
                bool tried = false, found = false;
                for (int it = 0; it < 4 && Pr < P; ++it, Pr += 4) {
                    tried = true;
                    if (fits(Pr) && hp.nP <= hr * unit) { P = Pr; found = true; break; }
                }
                if (tried && !found && !fits(P)) return "no patch size fits k_substep_pair";   // (the trials rebuilt hp: back to the size kept)
            }
        }
        out.pair_kernel = true;
    } else if (pair_nodes > 0) {
        P = pair_nodes;
        if (!build_patches2(m.t, m.ghost3, m.Nn, m.Ne, P, D, order, hp)) { snprintf(msg, sizeof msg, "multi-sub-step patch construction failed (pair_nodes=%d)", P); return msg; }
    } else {
        // as plan_patches: whole rounds of resident workgroups -- j workgroups per CU at a time, j = 1 first (a small mesh
        // is fastest with ONE workgroup on every CU: 10 km, 247 patches of 120 nodes 1.06 ms/step, 265 patches of 112 nodes 1.30)
        bool done = false;
        for (int j = 1; j <= (single_round_only ? 1 : 512) && !done; ++j) {
            P = (int)(((long long)m.Nn + (long long)j * cus - 1) / ((long long)j * cus));
            P = std::max(32, (P + 3) & ~3);
            if (P > 256) continue;
            if (!build_patches2(m.t, m.ghost3, m.Nn, m.Ne, P, D, order, hp)) continue;
            const size_t lds_cap = (j == 1 ? 160 : 80) * 1024;  // one workgroup per CU may take it all; otherwise two must fit
            done = multi_lds_of(hp) <= lds_cap && (hp.nP <= j * cus || P == 32);
        }
        if (!done) return single_round_only ? "the mesh does not fit one multi-sub-step patch per CU" : "no multi-sub-step patch size fits (node numbering without locality?)";
    }
    out.P = P;
    out.lds = out.pair_kernel ? pair_lds_of(hp) : multi_lds_of(hp);
    if (out.lds > 160 * 1024) { snprintf(msg, sizeof msg, "multi-sub-step patches need %zu B of LDS", out.lds); return msg; }
    // one patch per CU: 768 threads when a level does not fit 512 (10 km, D = 4: 0.98 -> 0.93 ms/step; 1 024 threads would force
    // 128 VGPRs + 40 spilled: 1.53); several patches per CU: 512, the outer levels take a second round of the block
    out.threads = out.pair_kernel ? pair_T : hp.EDmax <= 256 ? 256 : (hp.EDmax <= 512 || hp.nP > cus) ? 512 : 768;
    {   // NodalConnectivity rows in patch-local slots, for D smoother sweeps per launch (k_smooth_multi)
        const int Nn = m.Nn;
        out.pnbr.assign((size_t)hp.nP * W2 * hp.NSmax, 0xFFFF);
        std::vector<int> slot_of(Nn, -1);
        bool closed = (int)n2n_cnt.size() == Nn && n2n.size() == (size_t)W2 * Nn;
        for (int q = 0; q < hp.nP && closed; ++q) {
            const int *nd = hp.pnodes.data() + (size_t)q * hp.NDmax;
            const int nS = hp.ncnt[(size_t)q * (D + 1) + D - 1], nD = hp.ncnt[(size_t)q * (D + 1) + D];
            for (int i = 0; i < nD; ++i) slot_of[nd[i]] = i;
            for (int i = 0; i < nS && closed; ++i)
                for (int k = 0; k < n2n_cnt[nd[i]]; ++k) {
                    const int sl = slot_of[n2n[(size_t)k * Nn + nd[i]]];
                    if (sl < 0) { closed = false; break; }  // a caller-supplied row that reaches beyond the node's elements
                    out.pnbr[((size_t)q * W2 + k) * hp.NSmax + i] = (unsigned short)sl;
                }
            for (int i = 0; i < nD; ++i) slot_of[nd[i]] = -1;
        }
        if (!closed) out.pnbr.clear();
        out.smooth_lds = 4 * (size_t)hp.NDmax * sizeof(double) + (size_t)hp.NSmax;
    }
    return "";
}

// ------------------------------------------------------------------------------------------------
// The two-ring patches of k_substep_pair<HALO>: a rank of SEVERAL (own nodes first, ghosts behind them, one layer of ghost elements).  The patches are
// cut over the OWN nodes; a ghost node is never solved here (its fan is not complete on this rank), so
//   E_1 = every element touching an own node of the patch,        N_1 = their nodes (own nodes of other patches, ghosts),
//   E_2 = E_1 + every element touching an OWN node of N_1,        N_2 = their nodes,
// and the elements without any own node (they exist in a rank's mesh and the reference updates them) form patches of their own without nodes to solve.
// flags per patch: bit 0 = takes part in the exchange at all (a ghost among N_2, or it sends or receives), bit 1 = an own node of it is sent,
// bit 2 = a ghost in N_1 (it receives between the two sub-steps).  The patches come out band first (bits 1 | 2), then the rest of bit 0, then the interior.
struct PairHaloPlan {
    HostPatches2 hp;
    std::vector<unsigned char> pflags;
    int nG = 0, nBand = 0, P = 0;
    size_t lds = 0;
};

inline bool build_pair_patches_mr(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int No, int P, const std::vector<int> &order,
                                  const std::vector<char> &sent /*[No] != 0: the node is in a send list*/, PairHaloPlan &out) {
    if (P < 1 || No < 0 || No > Nn || (int)order.size() < No || (int)sent.size() < No) return false;
    const int D = 2;
    std::vector<int> off, adj;
    node_fans(t, Nn, Ne, off, adj);
    const int nNodePatches = (No + P - 1) / P;
    std::vector<int> patch_of(Nn, -1);
    for (int i = 0; i < No; ++i) patch_of[order[i]] = i / P;
    std::vector<int> writer(Ne, -1), orphans;
    for (int e = 0; e < Ne; ++e) {
        int wq = -1;
        for (int k = 0; k < 3; ++k) { const int q = patch_of[t[k][e]]; if (q >= 0 && (wq < 0 || q < wq)) wq = q; }
        writer[e] = wq;
        if (wq < 0) orphans.push_back(e);
    }
    const int EORPH = 1024;   // (two rounds of the 512-thread block: the limit of E_1)
    const int nOrph = ((int)orphans.size() + EORPH - 1) / EORPH, nP = nNodePatches + nOrph;
    for (size_t i = 0; i < orphans.size(); ++i) writer[orphans[i]] = nNodePatches + (int)(i / EORPH);

    std::vector<std::vector<int>> pel(nP), pnd(nP);
    std::vector<std::vector<unsigned short>> tri_l(nP);
    std::vector<std::vector<std::vector<unsigned short>>> fan_l(nP);
    std::vector<int> ncnt((size_t)nP * (D + 1), 0), ecnt((size_t)nP * D, 0);
    std::vector<unsigned char> flags(nP, 0);
    std::vector<int> emark(Ne, -1), eslot(Ne, -1), slot_of(Nn, -1);
    int NDmax = 0, NSmax = 0, EDmax = 0, ESmax = 0, Wp = 0;
    for (int q = 0; q < nP; ++q) {
        auto &nd = pnd[q];
        auto &el = pel[q];
        int *nc = ncnt.data() + (size_t)q * (D + 1), *ec = ecnt.data() + (size_t)q * D;
        if (q < nNodePatches) {
            const int a = q * P, bnd = std::min(No, a + P);
            for (int i = a; i < bnd; ++i) { slot_of[order[i]] = (int)nd.size(); nd.push_back(order[i]); if (sent[order[i]]) flags[q] |= 2; }
        }
        nc[0] = (int)nd.size();
        int n_prev = 0, e_prev = 0;
        for (int lev = 1; lev <= D; ++lev) {
            std::vector<int> add;
            if (q >= nNodePatches && lev == 1) {   // a patch of elements without an own node
                const int a = (q - nNodePatches) * EORPH, bnd = std::min((int)orphans.size(), a + EORPH);
                for (int i = a; i < bnd; ++i) { emark[orphans[i]] = q; add.push_back(orphans[i]); }
            } else {
                for (int i = n_prev; i < nc[lev - 1]; ++i) {
                    if (nd[i] >= No) continue;     // a ghost is not solved here: its other elements are not needed (and not all on this rank)
                    for (int j = off[nd[i]]; j < off[nd[i] + 1]; ++j) {
                        const int e = adj[j];
                        if (emark[e] != q) { emark[e] = q; add.push_back(e); }
                    }
                }
            }
            std::sort(add.begin(), add.end());
            el.insert(el.end(), add.begin(), add.end());
            ec[lev - 1] = (int)el.size();
            std::vector<int> addn;
            for (int l = e_prev; l < ec[lev - 1]; ++l)
                for (int k = 0; k < 3; ++k) {
                    const int n = t[k][el[l]];
                    if (slot_of[n] == -1) { slot_of[n] = -2; addn.push_back(n); }
                }
            std::sort(addn.begin(), addn.end());
            for (int n : addn) { slot_of[n] = (int)nd.size(); nd.push_back(n); if (n >= No) flags[q] |= (lev == 1 ? 5 : 1); }
            nc[lev] = (int)nd.size();
            n_prev = nc[lev - 1]; e_prev = ec[lev - 1];
        }
        if (flags[q] & 6) flags[q] |= 1;
        const bool too_big = nd.size() > 1024 || el.size() > 8191;
        if (!too_big) {
            for (size_t l = 0; l < el.size(); ++l) eslot[el[l]] = (int)l;
            auto &tl = tri_l[q];
            tl.assign(4 * el.size(), 0);
            for (size_t l = 0; l < el.size(); ++l)
                for (int k = 0; k < 3; ++k) tl[4 * l + k] = (unsigned short)slot_of[t[k][el[l]]];
            auto &fl = fan_l[q];
            fl.assign(nc[1], {});
            for (int i = 0; i < nc[1]; ++i) {
                const int n = nd[i];
                if (n >= No) continue;             // (never solved here: its row stays empty)
                for (int j = off[n]; j < off[n + 1]; ++j) {  // ascending element id = the order of the serial scatter
                    const int e = adj[j];
                    int k = 0;
                    while (t[k][e] != n) ++k;
                    fl[i].push_back((unsigned short)((eslot[e] << 3) | (ghost3[3 * (size_t)e + k] ? 4 : 0) | k));
                }
                Wp = std::max(Wp, (int)fl[i].size());
            }
        }
        for (int n : nd) slot_of[n] = -1;
        if (too_big) return false;
        NDmax = std::max(NDmax, nc[D]); NSmax = std::max(NSmax, nc[1]);
        EDmax = std::max(EDmax, ec[1]); ESmax = std::max(ESmax, ec[0]);
    }
    // band first, then the other patches that take part in the exchange, then the interior
    std::vector<int> perm;
    for (int q = 0; q < nP; ++q) if (flags[q] & 6) perm.push_back(q);
    const int nBand = (int)perm.size();
    for (int q = 0; q < nP; ++q) if ((flags[q] & 1) && !(flags[q] & 6)) perm.push_back(q);
    const int nG = (int)perm.size();
    for (int q = 0; q < nP; ++q) if (!(flags[q] & 1)) perm.push_back(q);

    HostPatches2 &hp = out.hp;
    hp = HostPatches2{};
    hp.nP = nP; hp.D = D;
    hp.NDmax = (NDmax + 1) & ~1; hp.NSmax = (NSmax + 1) & ~1; hp.EDmax = (EDmax + 1) & ~1; hp.ESmax = std::max(2, (ESmax + 1) & ~1); hp.Wp = std::max(Wp, 1);
    hp.ncnt.assign((size_t)nP * (D + 1), 0); hp.ecnt.assign((size_t)nP * D, 0);
    hp.pnodes.assign((size_t)nP * hp.NDmax, 0);
    hp.pelem.assign((size_t)nP * hp.EDmax, 0);
    hp.ptri.assign((size_t)nP * hp.EDmax * 4, 0);
    hp.pfan.assign((size_t)nP * hp.Wp * hp.NSmax, 0xFFFF);
    out.pflags.assign(nP, 0);
    for (int qn = 0; qn < nP; ++qn) {
        const int q = perm[qn];
        for (int i = 0; i <= D; ++i) hp.ncnt[(size_t)qn * (D + 1) + i] = ncnt[(size_t)q * (D + 1) + i];
        for (int i = 0; i < D; ++i) hp.ecnt[(size_t)qn * D + i] = ecnt[(size_t)q * D + i];
        out.pflags[qn] = flags[q];
        std::copy(pnd[q].begin(), pnd[q].end(), hp.pnodes.begin() + (size_t)qn * hp.NDmax);
        for (size_t l = 0; l < pel[q].size(); ++l) { const int e = pel[q][l]; hp.pelem[(size_t)qn * hp.EDmax + l] = (writer[e] == q) ? e : ~e; }
        std::copy(tri_l[q].begin(), tri_l[q].end(), hp.ptri.begin() + (size_t)qn * hp.EDmax * 4);
        for (size_t i = 0; i < fan_l[q].size(); ++i)
            for (size_t k = 0; k < fan_l[q][i].size(); ++k) hp.pfan[(size_t)qn * hp.Wp * hp.NSmax + k * hp.NSmax + i] = fan_l[q][i][k];
    }
    out.nG = nG; out.nBand = nBand; out.P = P;
    out.lds = pair_lds_of(hp);
    return true;
}

// which patch size: the largest whose patches fit k_substep_pair twice on a CU (as plan_patches2 does for one rank); `hint`: the size kept for the mesh before
inline std::string plan_pair_patches_mr(const MeshView &m, bool used_hilbert, int pair_nodes, int cus, const std::vector<char> &sent, PairHaloPlan &out, int hint = 0) {
    if (m.No <= 0) return "no own nodes";
    std::vector<int> order(m.No);
    for (int i = 0; i < m.No; ++i) order[i] = i;
    if (used_hilbert) hilbert_order(m.x0, m.y0, m.No, order);
    const size_t cap = 80 * 1024;
    auto own_max = [&]() { int v = 0; for (int q = 0; q < out.hp.nP; ++q) v = std::max(v, out.hp.ncnt[(size_t)q * 3]); return v; };
    auto fits = [&](int PP) { return build_pair_patches_mr(m.t, m.ghost3, m.Nn, m.Ne, m.No, PP, order, sent, out) && out.lds <= cap && pair_kernel_fits(out.hp, own_max()); };
    cus = std::max(cus, 1);
    if (pair_nodes > 0) return fits(pair_nodes) ? "" : "patches of that size do not fit k_substep_pair (80 KB of LDS, three rounds of elements)";
    const int hi0 = std::min(512, std::max(68, (int)((((long long)m.No + 2 * cus - 1) / (2 * cus) + 3) & ~3ll) + 4));
    int lo = 64, hi = hi0, P = 0;
    if (hint >= 64 && hint < hi0 && fits(hint)) { P = hint; hi = 0; }
    else if (hint >= 68 && hint < hi0) hi = hint;
    if (hi > 0) {
        if (!fits(lo)) return "no patch size fits k_substep_pair (node numbering without locality?)";
        int last_built = lo;
        while (hi - lo > 4) {
            const int mid = ((lo + hi) / 2 + 3) & ~3;
            if (mid >= hi) break;
            last_built = mid;
            if (fits(mid)) lo = mid; else hi = mid;
        }
        P = lo;
        if (last_built != P && !fits(P)) return "no patch size fits k_substep_pair";
    }
    // a whole round where the partition is ONE round of workgroups (as plan_patches2).  More: the smallest patches that stay under the multiple of HALF a round
    // (one workgroup per CU) the largest ones are under -- as plan_patches2, but 7 % short of it (the band patches wait for the neighbour ranks: the rounds are
    // less even).  Rank 0 of two of the 2 km mesh, 60 launches: 855 patches of 428 nodes 2.91 ms, 924-963 of 396-380 2.85-2.87, 1 016 of 360 (two whole rounds
    // but for eight) 3.01 (profiles/r05_experiments/r5_pairnodes.log)
    const int slots = 2 * cus, k = (out.hp.nP + slots - 1) / slots, P_fit = P;
    if (k == 1 && out.hp.nP > 0 && out.hp.nP != k * slots) {
        int Pr = std::max(64, (int)((((long long)m.No + (long long)k * slots - 1) / ((long long)k * slots) + 3) & ~3ll));
        bool tried = false;
        for (int it = 0; it < 6 && Pr < P; ++it, Pr += 4) { tried = true; if (fits(Pr) && out.hp.nP <= k * slots) { P = Pr; break; } }
        if (tried && P == P_fit && !fits(P)) return "no patch size fits k_substep_pair";
    } else if (k >= 2) {
        const int unit = cus, hr = (out.hp.nP + unit - 1) / unit;
        const long long target = (long long)(0.93 * (double)hr * unit);
        int Pr = std::max(64, (int)((((long long)m.No + target - 1) / std::max(target, 1ll) + 3) & ~3ll));
        bool tried = false, found = false;
        for (int it = 0; it < 4 && Pr < P; ++it, Pr += 4) { tried = true; if (fits(Pr) && out.hp.nP <= hr * unit) { P = Pr; found = true; break; } }
        if (tried && !found && !fits(P)) return "no patch size fits k_substep_pair";
    }
    return "";
}

// ------------------------------------------------------------------------------------------------
// Node-ring patches for the open-water smoother alone (k_smooth_multi on meshes that do not use k_substep_multi): patches of 256
// consecutive own nodes (or consecutive along the Hilbert curve the sub-step patches were cut along), D rings of neighbours through
// the NodalConnectivity rows.  Only the node levels and the rows in patch-local slots are filled in.
struct SmoothPlan {
    int nP = 0, D = 0, NDmax = 0, NSmax = 0, own_is_block = 0;
    size_t lds = 0;
    std::vector<int> ncnt, pnodes;
    std::vector<unsigned short> pnbr;
};

inline std::string plan_smooth_patches(const MeshView &m, bool used_hilbert, int D, const std::vector<int> &n2n, const std::vector<int> &n2n_cnt, int W2,
                                       SmoothPlan &out) {
    char msg[160];
    const int Nn = m.Nn, P = 256;
    if (D < 1) return "smoother patches: depth out of range";
    if (m.No != Nn || (int)n2n_cnt.size() != Nn || n2n.size() != (size_t)W2 * Nn) return "smoother patches need a single-rank mesh";
    std::vector<int> order(Nn);
    for (int i = 0; i < Nn; ++i) order[i] = i;
    if (used_hilbert) hilbert_order(m.x0, m.y0, Nn, order);
    const int nP = (Nn + P - 1) / P;
    std::vector<int> ncnt((size_t)nP * (D + 1), 0), slot_of(Nn, -1);
    std::vector<std::vector<int>> pnd(nP);
    int NDmax = 0, NSmax = 0;
    for (int q = 0; q < nP; ++q) {
        auto &nd = pnd[q];
        const int a = q * P, bnd = std::min(Nn, a + P);
        for (int i = a; i < bnd; ++i) { slot_of[order[i]] = (int)nd.size(); nd.push_back(order[i]); }
        int *nc = ncnt.data() + (size_t)q * (D + 1);
        nc[0] = bnd - a;
        int prev = 0;
        for (int lev = 1; lev <= D; ++lev) {
            std::vector<int> add;
            for (int i = prev; i < nc[lev - 1]; ++i) {
                const int n = nd[i];
                for (int k = 0; k < n2n_cnt[n]; ++k) {
                    const int nb = n2n[(size_t)k * Nn + n];
                    if (slot_of[nb] == -1) { slot_of[nb] = -2; add.push_back(nb); }
                }
            }
            std::sort(add.begin(), add.end());
            for (int n : add) { slot_of[n] = (int)nd.size(); nd.push_back(n); }
            prev = nc[lev - 1];
            nc[lev] = (int)nd.size();
        }
        for (int n : nd) slot_of[n] = -1;
        if (nd.size() > 65535) return "smoother patch too large";
        NDmax = std::max(NDmax, nc[D]); NSmax = std::max(NSmax, nc[D - 1]);
    }
    NDmax = (NDmax + 1) & ~1; NSmax = (NSmax + 1) & ~1;
    const size_t lds = 4 * (size_t)NDmax * sizeof(double) + (size_t)NSmax;
    if (lds > 64 * 1024) { snprintf(msg, sizeof msg, "smoother patches need %zu B of LDS (numbering without locality?)", lds); return msg; }
    out = SmoothPlan{};
    out.pnodes.assign((size_t)nP * NDmax, 0);
    out.pnbr.assign((size_t)nP * W2 * NSmax, 0xFFFF);
    for (int q = 0; q < nP; ++q) {
        const auto &nd = pnd[q];
        std::copy(nd.begin(), nd.end(), out.pnodes.begin() + (size_t)q * NDmax);
        for (size_t i = 0; i < nd.size(); ++i) slot_of[nd[i]] = (int)i;
        const int nS = ncnt[(size_t)q * (D + 1) + D - 1];
        for (int i = 0; i < nS; ++i)
            for (int k = 0; k < n2n_cnt[nd[i]]; ++k) out.pnbr[((size_t)q * W2 + k) * NSmax + i] = (unsigned short)slot_of[n2n[(size_t)k * Nn + nd[i]]];
        for (int n : nd) slot_of[n] = -1;
    }
    out.ncnt = std::move(ncnt);
    out.nP = nP; out.D = D; out.NDmax = NDmax; out.NSmax = NSmax; out.lds = lds;
    out.own_is_block = (P == NXS_CUT_BLOCK && !used_hilbert) ? 1 : 0;
    return "";
}

// ------------------------------------------------------------------------------------------------
// Tables of the halo exchange fused into the sub-step / smoother kernels (HaloFused): CSR of what every own node sends, where every
// ghost sits in a mailbox half, the patches reordered boundary-first.
struct HaloLists {  // the halo lists of nxs_dyn_set_halo
    const std::vector<int> *send_offsets, *recv_offsets;  // [ns + 1], [nr + 1]
    const std::vector<int> *send_index, *recv_index;      // own nodes sent / ghost nodes received, per neighbour segment
    int ns, nr;
};
struct HaloFusedPlan {
    std::vector<int> sptr, sk, spos, goff, gsrl, gk, send_block_rank;
    int n_send_blocks = 0, n_boundary = 0;
    bool reordered = false;  // hp was rewritten with the boundary patches first
};

inline std::string plan_halo_fused(int Nn, int No, const HaloLists &hl, HostPatches &hp, HaloFusedPlan &out) {
    const int nP = hp.nP, ns = hl.ns, nr = hl.nr;
    if ((int)hl.recv_index->size() != Nn - No) return "fused halo tables: patches / halo lists missing";
    const std::vector<int> &so = *hl.send_offsets, &ro = *hl.recv_offsets, &si = *hl.send_index, &ri = *hl.recv_index;
    out = HaloFusedPlan{};
    // sending side: CSR over own nodes
    std::vector<int> &sptr = out.sptr;
    sptr.assign((size_t)No + 1, 0);
    for (int k = 0; k < ns; ++k)
        for (int j = so[k]; j < so[k + 1]; ++j) sptr[si[j] + 1]++;
    for (int n = 0; n < No; ++n) sptr[n + 1] += sptr[n];
    out.sk.assign(std::max(sptr[No], 1), 0); out.spos.assign(std::max(sptr[No], 1), 0);
    std::vector<int> fill(sptr.begin(), sptr.end() - 1);
    for (int k = 0; k < ns; ++k)
        for (int j = so[k]; j < so[k + 1]; ++j) {
            const int q = fill[si[j]]++;
            out.sk[q] = k;
            out.spos[q] = j - so[k];
        }
    // receiving side: where each ghost node sits inside a mailbox half (layout of k_halo_pull)
    out.goff.assign(std::max(Nn - No, 1), 0); out.gsrl.assign(std::max(Nn - No, 1), 0); out.gk.assign(std::max(Nn - No, 1), 0);
    for (int k = 0; k < nr; ++k) {
        const int off = ro[k], srl = ro[k + 1] - off;
        for (int j = off; j < ro[k + 1]; ++j) {
            out.goff[ri[j] - No] = 2 * off + (j - off);
            out.gsrl[ri[j] - No] = srl;
            out.gk[ri[j] - No] = k;
        }
    }
    // boundary patches: send something or stage a ghost node.  The patch arrays are rewritten with those patches
    // FIRST, so that the grid starts with them and "boundary" is blk < n_boundary (no lookup on the critical path)
    std::vector<int> order_b, order_i;
    for (int q = 0; q < nP; ++q) {
        const int *nd = hp.pnodes.data() + (size_t)q * hp.Mmax;
        bool bnd = false;
        for (int i = 0; i < hp.node_cnt[q] && !bnd; ++i) {
            const int n = nd[i];
            bnd = (n >= No) || (i < hp.own_cnt[q] && sptr[n + 1] > sptr[n]);
        }
        (bnd ? order_b : order_i).push_back(q);
    }
    const int nb = (int)order_b.size();
    bool sorted = true;
    for (int q = 0; q < nb; ++q) sorted = sorted && order_b[q] == q;
    if (!sorted) {
        std::vector<int> order(order_b);
        order.insert(order.end(), order_i.begin(), order_i.end());
        HostPatches r = hp;
        for (int q = 0; q < nP; ++q) {
            const int o = order[q];
            r.own_cnt[q] = hp.own_cnt[o]; r.elem_cnt[q] = hp.elem_cnt[o]; r.node_cnt[q] = hp.node_cnt[o];
            std::copy_n(hp.pnodes.begin() + (size_t)o * hp.Mmax, hp.Mmax, r.pnodes.begin() + (size_t)q * hp.Mmax);
            std::copy_n(hp.pelem.begin() + (size_t)o * hp.Emax, hp.Emax, r.pelem.begin() + (size_t)q * hp.Emax);
            std::copy_n(hp.ptri.begin() + (size_t)o * hp.Emax * 4, (size_t)hp.Emax * 4, r.ptri.begin() + (size_t)q * hp.Emax * 4);
            std::copy_n(hp.pfan.begin() + (size_t)o * hp.Wp * hp.Pmax, (size_t)hp.Wp * hp.Pmax, r.pfan.begin() + (size_t)q * hp.Wp * hp.Pmax);
        }
        hp = std::move(r);
        out.reordered = true;
    }
    {   // k_smooth_halo runs BLOCK own nodes per block: which blocks store into a mailbox
        const int B = NXS_CUT_BLOCK, nblk = std::max(1, (No + B - 1) / B);
        out.send_block_rank.assign(nblk, -1);
        int cnt = 0;
        for (int b = 0; b < nblk; ++b)
            if (sptr[std::min(No, (b + 1) * B)] > sptr[std::min(No, b * B)]) out.send_block_rank[b] = cnt++;
        out.n_send_blocks = cnt;
    }
    out.n_boundary = nb;
    return "";
}

// ------------------------------------------------------------------------------------------------
// Tables of the resident sub-step kernel: which patches own each patch's halo nodes; with option resident_overlap the element lists with the
// interior elements first.  ok == false: this partition cannot run it (why says so); the caller then runs one kernel per sub-step.
struct ResidentPlan {
    bool ok = false;
    std::string why;
    std::vector<int> nbr, cnt;  // [nP][NXS_CUT_RES_NBR], [nP]
    int max_nbr = 0;
    // overlap variant: the patch lists with the interior elements first
    std::vector<int> rpelem, ecut;
    std::vector<unsigned short> rptri, rpfan;
    double early_fraction = 0.;
};

inline void plan_resident(const HostPatches &hp, int Nn, int No, bool multi_rank, int n_send_procs, bool overlap, ResidentPlan &out) {
    out = ResidentPlan{};
    const int nP = hp.nP;
    char msg[160];
    auto refuse = [&](const char *w) { out.ok = false; out.why = w; };
    const bool big = resident_is_big(hp);
    if (hp.Emax > 512 * NXS_CUT_RESB_EPT || hp.Pmax > 512 * NXS_CUT_RESB_NPT || (big && hp.Mmax > 1024)) {
        snprintf(msg, sizeof msg, "a patch holds %d elements / %d own nodes / %d staged nodes (at most %d / %d / 1024)", hp.Emax, hp.Pmax, hp.Mmax, 512 * NXS_CUT_RESB_EPT, 512 * NXS_CUT_RESB_NPT);
        return refuse(msg);
    }
    std::vector<int> owner(Nn, -1);
    for (int q = 0; q < nP; ++q)
        for (int i = 0; i < hp.own_cnt[q]; ++i) owner[hp.pnodes[(size_t)q * hp.Mmax + i]] = q;
    out.nbr.assign((size_t)nP * NXS_CUT_RES_NBR, -1); out.cnt.assign(nP, 0);
    std::vector<char> ghost_taken(std::max(Nn - No, 1), 0);
    for (int q = 0; q < nP; ++q)
        for (int i = hp.own_cnt[q]; i < hp.node_cnt[q]; ++i) {
            const int g = hp.pnodes[(size_t)q * hp.Mmax + i];
            if (g >= No) {  // a ghost node (several ranks): it comes from the mailbox, and every patch that stages it notes what arrived in the ghosts' ring
                if (!multi_rank) return refuse("a ghost node on a single-rank handle");
                ghost_taken[g - No] = 1;
                continue;
            }
            const int o = owner[g];
            if (o < 0 || o == q) return refuse("a staged node nobody solves");
            int *row = out.nbr.data() + (size_t)q * NXS_CUT_RES_NBR;
            bool have = false;
            for (int k = 0; k < out.cnt[q]; ++k) have = have || row[k] == o;
            if (have) continue;
            if (out.cnt[q] == NXS_CUT_RES_NBR) return refuse("a patch has more neighbour patches than the kernel can wait for");
            row[out.cnt[q]++] = o;
        }
    for (int g = No; g < Nn; ++g) if (!ghost_taken[g - No]) return refuse("a ghost node no patch stages: nobody would note its velocities");
    if (multi_rank && n_send_procs > NXS_CUT_RES_MAXNB) return refuse("more neighbour ranks than the kernel keeps mailbox addresses for");
    out.max_nbr = nP > 0 ? *std::max_element(out.cnt.begin(), out.cnt.end()) : 0;
    if (overlap) {
        // Interior elements first: an element none of whose corners is a halo node of its patch needs nothing from outside, so its next
        // update can be computed while the patch waits for the exchange.  A stable partition of every patch's element list (whole
        // wavefronts of interior elements only: ecut is a multiple of 64 -- of 512, whole slices of one element per thread, for the large patches of
        // k_substep_resident_big); the fan entries keep their (ascending global element) order and
        // only name the new slots, so the additions of the gather stay the reference's.
        out.rpelem.assign(hp.pelem.size(), 0); out.ecut.assign(nP, 0);
        out.rptri.assign(hp.ptri.size(), 0); out.rpfan.assign(hp.pfan.size(), 0xFFFF);
        std::vector<int> newslot(hp.Emax);
        long long tot_early = 0, tot_e = 0;
        for (int q = 0; q < nP; ++q) {
            const int nE = hp.elem_cnt[q], nO = hp.own_cnt[q];
            const unsigned short *tq = hp.ptri.data() + (size_t)q * hp.Emax * 4;
            auto interior = [&](int l) { return tq[4 * l] < nO && tq[4 * l + 1] < nO && tq[4 * l + 2] < nO; };
            int nint = 0;
            for (int l = 0; l < nE; ++l) nint += interior(l) ? 1 : 0;
            int a = 0, b2 = nint;
            for (int l = 0; l < nE; ++l) newslot[l] = interior(l) ? a++ : b2++;
            for (int l = nE; l < hp.Emax; ++l) newslot[l] = l;
            for (int l = 0; l < hp.Emax; ++l) {
                out.rpelem[(size_t)q * hp.Emax + newslot[l]] = hp.pelem[(size_t)q * hp.Emax + l];
                for (int k = 0; k < 4; ++k) out.rptri[((size_t)q * hp.Emax + newslot[l]) * 4 + k] = tq[4 * l + k];
            }
            for (size_t i = (size_t)q * hp.Wp * hp.Pmax; i < (size_t)(q + 1) * hp.Wp * hp.Pmax; ++i) {
                const unsigned short ent = hp.pfan[i];
                out.rpfan[i] = ent == 0xFFFF ? ent : (unsigned short)((newslot[ent >> 3] << 3) | (ent & 7));
            }
            out.ecut[q] = big ? (nint / 512) * 512 : (nint / 64) * 64;
            tot_early += out.ecut[q]; tot_e += nE;
        }
        out.early_fraction = (double)tot_early / (double)std::max(1ll, tot_e);
    }
    out.ok = true;
}

}  // namespace nxs_cut

#endif  // NXS_PATCHCUT_HPP
