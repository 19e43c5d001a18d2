// nxs_dyn_patches.inl -- host side: cutting the mesh into the node patches of the fused sub-step kernels (DevPatches, one ring of
// halo; DevPatches2, D rings) and uploading them.  Textually included by nxs_dyn.hip inside its anonymous namespace.
// ------------------------------------------------------------------------------------------------
// Host: node patches for the fused sub-step kernel (see DevPatches).

// order: owned nodes in the order they are cut into patches of P.
// Ecap > 0: a patch is closed early when one more own node would take it past Ecap elements (the resident kernel holds one element per
// thread; a partition whose own nodes are not contiguous along the numbering -- an RCB part of a Hilbert-numbered mesh -- otherwise has a few
// patches of two distant blobs with 1.5 times the elements of the others, and the whole round waits for them).
bool build_patches_from_order(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int No, int P,
                              const std::vector<int> &order, HostPatches &out, int Ecap = 0) {
    // node -> elements CSR
    std::vector<int> off(Nn + 1, 0);
    for (int k = 0; k < 3; ++k) for (int e = 0; e < Ne; ++e) off[t[k][e] + 1]++;
    for (int n = 0; n < Nn; ++n) off[n + 1] += off[n];
    std::vector<int> adj(off[Nn]), fill(off.begin(), off.end() - 1);
    for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) adj[fill[t[k][e]]++] = e;  // ascending e per node

    std::vector<int> pstart;  // patch q owns order[pstart[q] .. pstart[q + 1])
    if (Ecap <= 0) {
        for (int a = 0; a < No; a += P) pstart.push_back(a);
    } else {
        std::vector<int> seen(Ne, -1);
        int cnt_n = 0, cnt_e = 0, q = 0;
        if (No > 0) pstart.push_back(0);
        for (int i = 0; i < No; ++i) {
            const int n = order[i];
            int fresh = 0;
            for (int j = off[n]; j < off[n + 1]; ++j) fresh += seen[adj[j]] != q ? 1 : 0;
            if (cnt_n > 0 && (cnt_n == P || cnt_e + fresh > Ecap)) {  // close the patch before this node
                pstart.push_back(i);
                ++q; cnt_n = 0; cnt_e = 0;
                fresh = off[n + 1] - off[n];
            }
            for (int j = off[n]; j < off[n + 1]; ++j) if (seen[adj[j]] != q) { seen[adj[j]] = q; }
            ++cnt_n; cnt_e += fresh;
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

// owned nodes sorted along a Hilbert curve through their coordinates
void hilbert_order(const double *x0, const double *y0, int No, std::vector<int> &order) {
    order.resize(No);
    for (int i = 0; i < No; ++i) order[i] = i;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300;
    for (int n = 0; n < No; ++n) { xmin = std::min(xmin, x0[n]); xmax = std::max(xmax, x0[n]); ymin = std::min(ymin, y0[n]); ymax = std::max(ymax, y0[n]); }
    const double ext = std::max(xmax - xmin, ymax - ymin);
    const double sc = ext > 0. ? 65535. / ext : 0.;
    auto hilbert = [](unsigned x, unsigned y) {
        unsigned long long d = 0;
        for (unsigned s2 = 1u << 15; s2 > 0; s2 >>= 1) {
            const unsigned rx = (x & s2) ? 1u : 0u, ry = (y & s2) ? 1u : 0u;
            d += (unsigned long long)s2 * s2 * ((3u * rx) ^ ry);
            if (ry == 0) {
                if (rx == 1) { x = s2 - 1 - x; y = s2 - 1 - y; }
                const unsigned t2 = x; x = y; y = t2;
            }
        }
        return d;
    };
    std::vector<unsigned long long> key(No);
    for (int n = 0; n < No; ++n) key[n] = hilbert((unsigned)((x0[n] - xmin) * sc), (unsigned)((y0[n] - ymin) * sc));
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return key[a] < key[b2]; });
}

bool build_patches(const std::vector<int> t[3], const unsigned char *ghost3, const double *x0, const double *y0, int Nn, int Ne,
                   int No, int P, HostPatches &out, int Ecap = 0) {
    // 1st try: the caller's node numbering (keeps the patch's nodal accesses contiguous)
    std::vector<int> order(No);
    for (int i = 0; i < No; ++i) order[i] = i;
    bool ok = build_patches_from_order(t, ghost3, Nn, Ne, No, P, order, out, Ecap);
    if (ok && out.avg_elems_per_own_node <= 3.0) return true;
    // numbering without locality: cut patches along a Hilbert curve through the node coordinates
    // (consecutive runs of a Hilbert curve are compact blobs: small halos)
    hilbert_order(x0, y0, No, order);
    HostPatches alt;
    if (build_patches_from_order(t, ghost3, Nn, Ne, No, P, order, alt, Ecap) && (!ok || alt.avg_elems_per_own_node < out.avg_elems_per_own_node)) {
        out = std::move(alt);
        out.used_hilbert = true;
        return true;
    }
    return ok;
}

// Host: D-ring patches of k_substep_multi (DevPatches2); single rank (every node owned, no orphan elements).
bool build_patches2(const std::vector<int> t[3], const unsigned char *ghost3, int Nn, int Ne, int P, int D, const std::vector<int> &order, HostPatches2 &out) {
    std::vector<int> off(Nn + 1, 0);
    for (int k = 0; k < 3; ++k) for (int e = 0; e < Ne; ++e) off[t[k][e] + 1]++;
    for (int n = 0; n < Nn; ++n) off[n + 1] += off[n];
    std::vector<int> adj(off[Nn]), fill(off.begin(), off.end() - 1);
    for (int e = 0; e < Ne; ++e) for (int k = 0; k < 3; ++k) adj[fill[t[k][e]]++] = e;  // ascending e per node
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
        if (nd.size() > 65535 || el.size() > 8191) return false;
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

int upload_patches2(nxs_dyn_handle *h, int D, bool single_round_only) {
    free_pool(h->pair_allocs);
    h->dpch2 = DevPatches2{};
    h->pair_ready = false;
    const DevMesh &m = h->dm;
    if (m.No != m.Nn) return fail(h, NXS_ERR_STATE, "multi-sub-step patches need a single-rank mesh");
    std::vector<int> order(m.Nn);
    for (int i = 0; i < m.Nn; ++i) order[i] = i;
    // the caller's numbering if it has locality, else the Hilbert curve the single-ring patches were cut along
    if (h->hp && h->hp->used_hilbert) hilbert_order(h->h_x0.data(), h->h_y0.data(), m.Nn, order);
    HostPatches2 hp;
    auto lds_of = [](const HostPatches2 &x) { return (4 * (size_t)x.NDmax + 6 * (size_t)x.EDmax + 2 + 4 * (size_t)x.ESmax) * sizeof(double); };
    int P = 0, threads = 512;
    if (h->pair_nodes > 0) {
        P = h->pair_nodes;
        if (!build_patches2(h->h_t, h->h_ghost.data(), m.Nn, m.Ne, P, D, order, hp)) return fail(h, NXS_ERR_INVALID, "multi-sub-step patch construction failed (pair_nodes=%d)", P);
    } else {
        // as upload_patches: whole rounds of resident workgroups -- j workgroups per CU at a time, j = 1 first (a small mesh
        // is fastest with ONE workgroup on every CU: 10 km, 247 patches of 120 nodes 1.06 ms/step, 265 patches of 112 nodes
        // 1.30)
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        cus = std::max(cus, 1);
        bool done = false;
        for (int j = 1; j <= (single_round_only ? 1 : 512) && !done; ++j) {
            P = (int)(((long long)m.Nn + (long long)j * cus - 1) / ((long long)j * cus));
            P = std::max(32, (P + 3) & ~3);
            if (P > 256) continue;
            if (!build_patches2(h->h_t, h->h_ghost.data(), m.Nn, m.Ne, P, D, order, hp)) continue;
            const size_t lds_cap = (j == 1 ? 160 : 80) * 1024;  // one workgroup per CU may take it all; otherwise two must fit
            done = lds_of(hp) <= lds_cap && (hp.nP <= j * cus || P == 32);
        }
        if (!done) return fail(h, NXS_ERR_INVALID, single_round_only ? "the mesh does not fit one multi-sub-step patch per CU" : "no multi-sub-step patch size fits (node numbering without locality?)");
    }
    h->pair_lds = lds_of(hp);
    if (h->pair_lds > 160 * 1024) return fail(h, NXS_ERR_INVALID, "multi-sub-step patches need %zu B of LDS", h->pair_lds);
    {   // one patch per CU: 768 threads when a level does not fit 512 (10 km, D = 4: 0.98 -> 0.93 ms/step; 1 024 threads would force
        // 128 VGPRs + 40 spilled: 1.53); several patches per CU: 512, the outer levels take a second round of the block
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        threads = hp.EDmax <= 256 ? 256 : (hp.EDmax <= 512 || hp.nP > cus) ? 512 : 768;
    }
    h->pair_threads = threads;
    if (getenv("NXS_DEBUG_PATCHES")) {
        std::vector<double> se(D, 0.), sn(D + 1, 0.);
        for (int q = 0; q < hp.nP; ++q) { for (int i = 0; i < D; ++i) se[i] += hp.ecnt[(size_t)q * D + i]; for (int i = 0; i <= D; ++i) sn[i] += hp.ncnt[(size_t)q * (D + 1) + i]; }
        fprintf(stderr, "[nxs] multi patches: D=%d P=%d nP=%d EDmax=%d ESmax=%d NDmax=%d NSmax=%d Wp=%d lds=%zu B threads=%d; elements per level x", D, P, hp.nP, hp.EDmax, hp.ESmax,
                hp.NDmax, hp.NSmax, hp.Wp, h->pair_lds, threads);
        for (int i = 0; i < D; ++i) fprintf(stderr, " %.3f", se[i] / std::max(m.Ne, 1));
        fprintf(stderr, "; nodes per level x");
        for (int i = 0; i <= D; ++i) fprintf(stderr, " %.3f", sn[i] / std::max(m.Nn, 1));
        fprintf(stderr, "\n");
    }
    DevPatches2 &d = h->dpch2;
    d.nP = hp.nP; d.D = D; d.NDmax = hp.NDmax; d.NSmax = hp.NSmax; d.EDmax = hp.EDmax; d.ESmax = hp.ESmax; d.Wp = hp.Wp;
    int rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ncnt, hp.ncnt))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ecnt, hp.ecnt))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pnodes, hp.pnodes))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pelem, hp.pelem))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ptri, hp.ptri))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pfan, hp.pfan))) return rc;
    {   // NodalConnectivity rows in patch-local slots, for D smoother sweeps per launch (k_smooth_multi)
        const int W2 = m.W2, Nn = m.Nn;
        std::vector<unsigned short> pnbr((size_t)hp.nP * W2 * hp.NSmax, 0xFFFF);
        std::vector<int> slot_of(Nn, -1);
        bool closed = (int)h->h_n2n_cnt.size() == Nn && h->h_n2n.size() == (size_t)W2 * Nn;
        for (int q = 0; q < hp.nP && closed; ++q) {
            const int *nd = hp.pnodes.data() + (size_t)q * hp.NDmax;
            const int nS = hp.ncnt[(size_t)q * (D + 1) + D - 1], nD = hp.ncnt[(size_t)q * (D + 1) + D];
            for (int i = 0; i < nD; ++i) slot_of[nd[i]] = i;
            for (int i = 0; i < nS && closed; ++i)
                for (int k = 0; k < h->h_n2n_cnt[nd[i]]; ++k) {
                    const int sl = slot_of[h->h_n2n[(size_t)k * Nn + nd[i]]];
                    if (sl < 0) { closed = false; break; }  // a caller-supplied row that reaches beyond the node's elements
                    pnbr[((size_t)q * W2 + k) * hp.NSmax + i] = (unsigned short)sl;
                }
            for (int i = 0; i < nD; ++i) slot_of[nd[i]] = -1;
        }
        d.W2 = W2;
        d.pnbr = nullptr;
        if (closed && (rc = dev_upload(h, h->pair_allocs, &d.pnbr, pnbr))) return rc;
        h->smooth_lds = 4 * (size_t)hp.NDmax * sizeof(double) + (size_t)hp.NSmax;
    }
    // per-step constants of the multi kernel as one record per element / node (written by the prep kernels while these exist):
    // two base pointers instead of fourteen, half the load instructions
    h->pair_ready = true;
    h->pair_depth_built = D;
    return NXS_OK;
}

int upload_host_patches(nxs_dyn_handle *h, const HostPatches &hp) {
    free_pool(h->patch_allocs);
    h->res_ready = false; h->res_failed = false;  // (its tables lived in this pool)
    DevPatches &d = h->dpch;
    d = DevPatches{};
    d.nP = hp.nP; d.Pmax = hp.Pmax; d.Emax = hp.Emax; d.Mmax = hp.Mmax; d.Wp = hp.Wp;
    int rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.own_cnt, hp.own_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.elem_cnt, hp.elem_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.node_cnt, hp.node_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pnodes, hp.pnodes))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pelem, hp.pelem))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.ptri, hp.ptri))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pfan, hp.pfan))) return rc;
    if (hp.Mmax > 1024) return fail(h, NXS_ERR_INVALID, "a patch stages %d nodes (at most 1024: choose smaller patches)", hp.Mmax);
    std::vector<int2> pet((size_t)hp.nP * hp.Emax);
    for (size_t i = 0; i < pet.size(); ++i)
        pet[i] = make_int2(hp.pelem[i], (int)hp.ptri[4 * i] | ((int)hp.ptri[4 * i + 1] << 10) | ((int)hp.ptri[4 * i + 2] << 20));
    if ((rc = dev_upload(h, h->patch_allocs, &d.pet, pet))) return rc;
    return NXS_OK;
}

int upload_patches(nxs_dyn_handle *h) {
    free_pool(h->patch_allocs);
    h->res_ready = false; h->res_failed = false;
    h->dpch = DevPatches{};
    h->fused_lds = 0;
    const DevMesh &m = h->dm;
    const bool automatic = h->patch_nodes <= 0;
    HostPatches hp;
    int P = 0;
    auto build = [&](int PP) -> bool {
        // patches of up to ~200 nodes hold one element per thread of a 512-thread workgroup (the resident loop requires it, and one round
        // of the one-launch-per-sub-step kernel is as slow as its largest patch): none may exceed 480 elements
        if (!build_patches(h->h_t, h->h_ghost.data(), h->h_x0.data(), h->h_y0.data(), m.Nn, m.Ne, m.No, PP, hp, (PP > NXS_T256_MAXP && PP <= 208) ? 480 : 0)) return false;
        h->fused_lds = (4 * (size_t)hp.Mmax + 6 * (size_t)hp.Emax + 2) * sizeof(double);  // staged nodes, corner forces + their pair of zeros
        return true;
    };
    if (!automatic) {
        P = std::max(64, std::min(h->patch_nodes, 1024));
        for (;;) {
            if (!build(P)) return fail(h, NXS_ERR_INVALID, "patch construction failed (patch_nodes=%d)", P);
            if ((h->fused_lds <= 80 * 1024 && hp.Mmax <= 1024) || P <= 64) break;
            P = std::max(64, P * 3 / 4);
        }
    } else {
        // Large patches recompute few halo elements; the limits are the LDS of two resident workgroups per CU
        // (160 KiB / 2) and, above all, WHOLE ROUNDS: the grid runs in rounds of `slots` resident workgroups and a
        // last round that is partly empty costs as much as a full one.  So: the smallest number of rounds k whose
        // patch size ceil(No / (k*slots)) fits, e.g. 730 k nodes -> 3 rounds of 512 patches of 476 nodes (not 2.79
        // rounds of 512-node patches); 92 k nodes (one rank of eight) -> one round of 511 patches of 180 nodes.
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
        cus = std::max(cus, 1);
        const int slots512 = 2 * cus, slots256 = 4 * cus;  // 16 waves per CU (112 VGPRs): 2 x 512 or 4 x 256 threads
        bool done = false;
        if (h->fused == 4 && m.No > 0) {
            // the resident sub-step loop was asked for (before set_mesh): ONE round of 512-thread workgroups with one element per thread --
            // two patches per CU, or one of twice the size where those would be smaller than ~100 nodes (10 km, 30 k nodes: 255 patches of
            // 116 nodes 0.625 ms/step, 462 of 64 nodes 0.653) -- also where the one-launch-per-sub-step kernel would take smaller patches
            // (65 k - 90 k nodes: 256-thread workgroups, four per CU)
            int Pr = (int)(((long long)m.No + 2 * cus - 1) / (2 * cus));
            if (Pr < 100) Pr = (int)(((long long)m.No + cus - 1) / cus);
            Pr = std::max(32, (Pr + 3) & ~3);
            for (int it = 0; it < 4 && Pr <= 208 && !done; ++it, Pr += 4) {  // orphan patches (multi-rank) may add a few workgroups
                if (!build(Pr)) break;
                done = hp.Emax <= 512 && hp.nP <= 2 * cus && h->fused_lds <= 80 * 1024;
                if (done) P = Pr;
            }
        }
        for (int k = 1; k <= 64 && !done; ++k) {
            P = (int)(((long long)m.No + (long long)k * slots512 - 1) / ((long long)k * slots512));
            P = (P + 3) & ~3;
            if (P > 512) continue;
            if (P <= NXS_T256_MAXP) break;  // small mesh: the 256-thread kernel below
            for (int it = 0; it < 4 && !done; ++it) {  // orphan patches (multi-rank) may add a few workgroups
                if (it > 0) P += 4;
                if (!build(P)) return fail(h, NXS_ERR_INVALID, "patch construction failed (patch_nodes=%d)", P);
                if (h->fused_lds > 80 * 1024 || hp.Mmax > 1024) break;  // does not fit twice (or its corner slots do not fit 10 bits): more rounds of smaller patches
                done = hp.nP <= k * slots512;
            }
        }
        if (!done) {
            P = (int)(((long long)m.No + slots256 - 1) / slots256);
            P = std::max(64, std::min((P + 3) & ~3, NXS_T256_MAXP));
            if (!build(P)) return fail(h, NXS_ERR_INVALID, "patch construction failed (patch_nodes=%d)", P);
        }
    }
    if (h->fused_lds > 160 * 1024) return fail(h, NXS_ERR_INVALID, "patches need %zu B of LDS", h->fused_lds);
    if (getenv("NXS_DEBUG_PATCHES")) {
        long long se = 0, sm = 0;
        for (int q = 0; q < hp.nP; ++q) { se += hp.elem_cnt[q]; sm += hp.node_cnt[q]; }
        fprintf(stderr, "[nxs] patches: P=%d nP=%d Pmax=%d Emax=%d Mmax=%d Wp=%d avgE=%.1f avgM=%.1f lds=%zu B elems x%.3f\n", P, hp.nP, hp.Pmax,
                hp.Emax, hp.Mmax, hp.Wp, (double)se / hp.nP, (double)sm / hp.nP, h->fused_lds, (double)se / std::max(m.Ne, 1));
    }
    h->hf_ready = false;
    h->hp = std::make_shared<HostPatches>(std::move(hp));
    return upload_host_patches(h, *h->hp);
}


// Host: node-ring patches for the open-water smoother alone (k_smooth_multi on meshes that do not use k_substep_multi): patches of 256
// consecutive own nodes (or consecutive along the Hilbert curve the sub-step patches were cut along), D rings of neighbours through
// the NodalConnectivity rows.  Only the node levels and the rows in patch-local slots are filled in.
int build_smooth_patches(nxs_dyn_handle *h, int D) {
    free_pool(h->sm_allocs);
    h->dsm = DevPatches2{};
    h->sm_ready = false;
    const DevMesh &m = h->dm;
    const int Nn = m.Nn, W2 = m.W2, P = 256;
    if (m.No != Nn || (int)h->h_n2n_cnt.size() != Nn || h->h_n2n.size() != (size_t)W2 * Nn) return fail(h, NXS_ERR_STATE, "smoother patches need a single-rank mesh");
    std::vector<int> order(Nn);
    for (int i = 0; i < Nn; ++i) order[i] = i;
    if (h->hp && h->hp->used_hilbert) hilbert_order(h->h_x0.data(), h->h_y0.data(), Nn, order);
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
                for (int k = 0; k < h->h_n2n_cnt[n]; ++k) {
                    const int nb = h->h_n2n[(size_t)k * Nn + n];
                    if (slot_of[nb] == -1) { slot_of[nb] = -2; add.push_back(nb); }
                }
            }
            std::sort(add.begin(), add.end());
            for (int n : add) { slot_of[n] = (int)nd.size(); nd.push_back(n); }
            prev = nc[lev - 1];
            nc[lev] = (int)nd.size();
        }
        for (int n : nd) slot_of[n] = -1;
        if (nd.size() > 65535) return fail(h, NXS_ERR_INVALID, "smoother patch too large");
        NDmax = std::max(NDmax, nc[D]); NSmax = std::max(NSmax, nc[D - 1]);
    }
    NDmax = (NDmax + 1) & ~1; NSmax = (NSmax + 1) & ~1;
    const size_t lds = 4 * (size_t)NDmax * sizeof(double) + (size_t)NSmax;
    if (lds > 64 * 1024) return fail(h, NXS_ERR_INVALID, "smoother patches need %zu B of LDS (numbering without locality?)", lds);
    std::vector<int> pnodes((size_t)nP * NDmax, 0);
    std::vector<unsigned short> pnbr((size_t)nP * W2 * NSmax, 0xFFFF);
    for (int q = 0; q < nP; ++q) {
        const auto &nd = pnd[q];
        std::copy(nd.begin(), nd.end(), pnodes.begin() + (size_t)q * NDmax);
        for (size_t i = 0; i < nd.size(); ++i) slot_of[nd[i]] = (int)i;
        const int nS = ncnt[(size_t)q * (D + 1) + D - 1];
        for (int i = 0; i < nS; ++i)
            for (int k = 0; k < h->h_n2n_cnt[nd[i]]; ++k) pnbr[((size_t)q * W2 + k) * NSmax + i] = (unsigned short)slot_of[h->h_n2n[(size_t)k * Nn + nd[i]]];
        for (int n : nd) slot_of[n] = -1;
    }
    DevPatches2 &d = h->dsm;
    d.nP = nP; d.D = D; d.NDmax = NDmax; d.NSmax = NSmax; d.W2 = W2;
    d.own_is_block = (P == BLOCK && !(h->hp && h->hp->used_hilbert)) ? 1 : 0;
    int rc;
    if ((rc = dev_upload(h, h->sm_allocs, &d.ncnt, ncnt))) return rc;
    if ((rc = dev_upload(h, h->sm_allocs, &d.pnodes, pnodes))) return rc;
    if ((rc = dev_upload(h, h->sm_allocs, &d.pnbr, pnbr))) return rc;
    h->sm_lds = lds;
    h->sm_ready = true;
    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] smoother patches: D=%d nP=%d NDmax=%d NSmax=%d lds=%zu B\n", D, nP, NDmax, NSmax, lds);
    return NXS_OK;
}
