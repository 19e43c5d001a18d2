// nxs_dyn_patches.inl -- uploading what nxs_patchcut.hpp cuts: the node patches of the fused sub-step kernels (DevPatches, one ring of
// halo; DevPatches2, D rings) and of the smoother.  Textually included by nxs_dyn.hip inside its anonymous namespace.  All the table
// construction is host-only code in nxs_patchcut.hpp (compiled a second time under ASan / UBSan by tests/test_sanitizers.py); here are
// only the device allocations and copies.
// ------------------------------------------------------------------------------------------------

nxs_cut::MeshView mesh_view(const nxs_dyn_handle *h) {
    return nxs_cut::MeshView{h->h_t, h->h_ghost.data(), h->h_x0.data(), h->h_y0.data(), h->dm.Nn, h->dm.Ne, h->dm.No,
                             (h->have_halo && (int)h->h_sent.size() == h->dm.No && h->dm.No > 0) ? h->h_sent.data() : nullptr};
}

int device_cus(const nxs_dyn_handle *h) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess) { (void)hipGetLastError(); cus = 256; }
    return std::max(cus, 1);
}

int upload_patches2(nxs_dyn_handle *h, int D, bool single_round_only, bool for_pair_kernel = false) {
    drop_pool(h, h->pair_allocs);
    h->dpch2 = DevPatches2{};
    h->pair_ready = false;
    const DevMesh &m = h->dm;
    nxs_cut::Patch2Plan plan;
    const std::string why = nxs_cut::plan_patches2(mesh_view(h), (h->hp && h->hp->used_hilbert) || h->pair_hilbert == 1, h->pair_nodes, D, single_round_only, device_cus(h),
                                                   h->h_n2n, h->h_n2n_cnt, m.W2, plan, for_pair_kernel, h->pair_hint, h->pair_T);
    if (!why.empty()) return fail(h, m.No != m.Nn ? NXS_ERR_STATE : NXS_ERR_INVALID, "%s", why.c_str());
    const HostPatches2 &hp = plan.hp;
    h->pair_lds = plan.lds;
    h->pair_threads = plan.threads;
    h->pair_kernel = plan.pair_kernel;
    if (plan.pair_kernel && h->pair_nodes == 0) h->pair_hint = plan.P_fit;
    h->pair_own_max = 0;
    for (int q = 0; q < hp.nP; ++q) h->pair_own_max = std::max(h->pair_own_max, hp.ncnt[(size_t)q * (D + 1)]);
    if (getenv("NXS_DEBUG_PATCHES")) {
        std::vector<double> se(D, 0.), sn(D + 1, 0.);
        for (int q = 0; q < hp.nP; ++q) { for (int i = 0; i < D; ++i) se[i] += hp.ecnt[(size_t)q * D + i]; for (int i = 0; i <= D; ++i) sn[i] += hp.ncnt[(size_t)q * (D + 1) + i]; }
        fprintf(stderr, "[nxs] multi patches: D=%d P=%d nP=%d EDmax=%d ESmax=%d NDmax=%d NSmax=%d Wp=%d lds=%zu B threads=%d; elements per level x", D, plan.P, hp.nP, hp.EDmax, hp.ESmax,
                hp.NDmax, hp.NSmax, hp.Wp, h->pair_lds, plan.threads);
        for (int i = 0; i < D; ++i) fprintf(stderr, " %.3f", se[i] / std::max(m.Ne, 1));
        fprintf(stderr, "; nodes per level x");
        for (int i = 0; i <= D; ++i) fprintf(stderr, " %.3f", sn[i] / std::max(m.Nn, 1));
        fprintf(stderr, "\n");
    }
    {   // sums over the tables, for nxs_dyn_get_traffic_model
        auto &S2 = h->sums2;
        S2 = nxs_dyn_handle::PatchSums2{};
        S2.nP = hp.nP; S2.N.assign(D + 1, 0.); S2.E.assign(D, 0.);
        for (int q = 0; q < hp.nP; ++q) {
            for (int i = 0; i <= D; ++i) S2.N[i] += hp.ncnt[(size_t)q * (D + 1) + i];
            for (int i = 0; i < D; ++i) S2.E[i] += hp.ecnt[(size_t)q * D + i];
            S2.E1_second_round += std::max(0, hp.ecnt[(size_t)q * D] - 512);
            for (int l = 0; l < hp.ecnt[(size_t)q * D]; ++l) S2.W += hp.pelem[(size_t)q * hp.EDmax + l] >= 0 ? 1. : 0.;
        }
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
    d.pet = nullptr;
    if (plan.pair_kernel) {   // {element, corner slots in ten bits each}: 8 bytes per element instead of 12 (pair_kernel_fits: at most 1024 staged nodes)
        std::vector<int> pet(2 * hp.pelem.size());
        for (size_t i = 0; i < hp.pelem.size(); ++i) {
            pet[2 * i] = hp.pelem[i];
            pet[2 * i + 1] = (int)hp.ptri[4 * i] | ((int)hp.ptri[4 * i + 1] << 10) | ((int)hp.ptri[4 * i + 2] << 20);
        }
        const int *dpet = nullptr;
        if ((rc = dev_upload(h, h->pair_allocs, &dpet, pet))) return rc;
        d.pet = reinterpret_cast<const int2 *>(dpet);
    }
    d.pfan8 = nullptr;
    {   // the first eight fan entries of every solved node, decoded: the LDS index of the corner's force, two per word (nxs_cut::decode_fan8)
        std::vector<unsigned int> f8;
        nxs_cut::decode_fan8(hp, f8);
        if ((rc = dev_upload(h, h->pair_allocs, &d.pfan8, f8))) return rc;
    }
    // NodalConnectivity rows in patch-local slots, for D smoother sweeps per launch (k_smooth_multi)
    d.W2 = m.W2;
    d.pnbr = nullptr;
    if (!plan.pnbr.empty() && (rc = dev_upload(h, h->pair_allocs, &d.pnbr, plan.pnbr))) return rc;
    h->smooth_lds = plan.smooth_lds;
    h->pair_ready = true;
    h->pair_depth_built = D;
    // the same patches as ONE data-flow launch per step (k_substep_flow): what each patch waits for, the queues, the counters
    h->flow = PairFlow{};
    h->flow_ready = false;
    if (plan.pair_kernel && plan.threads == 512 && flow_wanted(h) && !h->flow_failed) {
        std::vector<int> ptr, dep;
        auto refuse = [&](const char *w) { if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] one data-flow launch per step not possible: %s\n", w); return NXS_OK; };
        if (!nxs_cut::build_flow_deps(hp, m.Nn, m.Ne, ptr, dep)) return refuse("a node without an owner or an element without a writer among the patches");
        if (32ull * (unsigned long long)m.Ne >= (1ull << 32)) return refuse("the element state does not fit a 32-bit buffer offset");
        PairFlow &f = h->flow;
        if ((rc = dev_upload(h, h->pair_allocs, &f.dep_ptr, ptr))) return rc;
        if ((rc = dev_upload(h, h->pair_allocs, &f.dep, dep))) return rc;
        h->flow_words = 8 * 32 + 16 * (size_t)hp.nP;
        if ((rc = dev_alloc(h, h->pair_allocs, &f.queue, h->flow_words))) return rc;
        f.done = f.queue + 8 * 32;
        if ((rc = dev_alloc(h, h->pair_allocs, &f.error, 1))) return rc;
        HIPCHK(h, hipMemsetAsync(f.error, 0, sizeof(int), h->stream));
        nxs_cut::flow_queues(hp.nP, f.qstart);
        const void *kern = flow_kernel(h);
        HIPCHK(h, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->pair_lds));
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, h->pair_lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return refuse("the kernel does not fit a CU"); }
        // as many workgroups as the device holds at once (a multiple of eight: every queue gets the same number), never more than there are patches
        h->flow_grid = std::max(8, std::min(per_cu * device_cus(h), hp.nP) & ~7);
        int max_dep = 0;
        for (int q = 0; q < hp.nP; ++q) max_dep = std::max(max_dep, ptr[q + 1] - ptr[q]);
        if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] data-flow launch: %d workgroups (%d per CU) over %d patches in 8 queues, a patch waits for %.1f patches (at most %d)\n",
                                                 h->flow_grid, per_cu, hp.nP, (double)dep.size() / hp.nP, max_dep);
        h->flow_ready = true;
    }
    return NXS_OK;
}

// Several ranks: the two-ring patches of k_substep_pair<HALO> over this rank's own nodes (nxs_cut::plan_pair_patches_mr), their duties in the exchange, the
// ticket words -- and the claim on the device's workgroup slots for the patches that wait for a neighbour rank INSIDE the launch (they all have to be on
// a CU at once; ranks that share a device share its slots: nxs_resident_registry.hpp).  NXS_OK with pair_ready == false: not possible here.
int upload_pair_patches_mr(nxs_dyn_handle *h) {
    drop_pool(h, h->pair_allocs);
    h->dpch2 = DevPatches2{};
    h->pair_ready = false;
    h->pairh = PairHalo{};
    if (h->pair_claim) { resident_registry_release(h, nxs_reg::KIND_PAIR); h->pair_claim = false; }
    const DevMesh &m = h->dm;
    std::vector<char> sent((size_t)std::max(m.No, 1), 0);
    for (int n : h->h_send_index) if (n >= 0 && n < m.No) sent[n] = 1;
    nxs_cut::PairHaloPlan plan;
    const int cus = device_cus(h);
    const std::string why = nxs_cut::plan_pair_patches_mr(mesh_view(h), h->hp && h->hp->used_hilbert, h->pair_nodes, cus, sent, plan, h->pair_hint);
    auto refuse = [&](const char *w) {
        if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] rank %d: two sub-steps per launch not possible: %s\n", h->rank, w);
        h->pair_failed = true;
        return NXS_OK;
    };
    if (!why.empty()) return refuse(why.c_str());
    const HostPatches2 &hp = plan.hp;
    // every patch that takes part in the exchange may wait for a neighbour rank inside the launch: all of them resident at once, with room to spare
    if (plan.nG > cus) return refuse("more patches along the partition boundary than half the device's workgroup slots");
    {
        std::string w2;
        if (plan.nG > 0 && !resident_registry_claim(h, plan.nG, 2 * cus, &w2, nxs_reg::KIND_PAIR)) return refuse(w2.c_str());
        if (plan.nG > 0 && h->res_ready) { h->res_ready = false; release_graph(h); }   // (a handle holds ONE claim: the resident loop's went with this one)
        h->pair_claim = plan.nG > 0;
    }
    if (h->pair_nodes == 0) h->pair_hint = plan.P;
    h->pair_lds = plan.lds; h->pair_threads = 512; h->pair_kernel = true;
    h->pair_own_max = 0;
    for (int q = 0; q < hp.nP; ++q) h->pair_own_max = std::max(h->pair_own_max, hp.ncnt[(size_t)q * 3]);
    {
        auto &S2 = h->sums2;
        S2 = nxs_dyn_handle::PatchSums2{};
        S2.nP = hp.nP; S2.N.assign(3, 0.); S2.E.assign(2, 0.);
        for (int q = 0; q < hp.nP; ++q) {
            for (int i = 0; i <= 2; ++i) S2.N[i] += hp.ncnt[(size_t)q * 3 + i];
            for (int i = 0; i < 2; ++i) S2.E[i] += hp.ecnt[(size_t)q * 2 + i];
            S2.E1_second_round += std::max(0, hp.ecnt[(size_t)q * 2] - 512);
            for (int l = 0; l < hp.ecnt[(size_t)q * 2]; ++l) S2.W += hp.pelem[(size_t)q * hp.EDmax + l] >= 0 ? 1. : 0.;
        }
    }
    if (getenv("NXS_DEBUG_PATCHES"))
        fprintf(stderr, "[nxs] rank %d pair patches (several ranks): P=%d nP=%d (%d in the exchange, %d of them band) EDmax=%d ESmax=%d NDmax=%d NSmax=%d lds=%zu B; elements x %.3f / %.3f, nodes x %.3f / %.3f\n",
                h->rank, plan.P, hp.nP, plan.nG, plan.nBand, hp.EDmax, hp.ESmax, hp.NDmax, hp.NSmax, plan.lds, h->sums2.E[0] / std::max(m.Ne, 1), h->sums2.E[1] / std::max(m.Ne, 1),
                h->sums2.N[1] / std::max(m.No, 1), h->sums2.N[2] / std::max(m.No, 1));
    DevPatches2 &d = h->dpch2;
    d.nP = hp.nP; d.D = 2; d.NDmax = hp.NDmax; d.NSmax = hp.NSmax; d.EDmax = hp.EDmax; d.ESmax = hp.ESmax; d.Wp = hp.Wp;
    int rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ncnt, hp.ncnt))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ecnt, hp.ecnt))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pnodes, hp.pnodes))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pelem, hp.pelem))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.ptri, hp.ptri))) return rc;
    if ((rc = dev_upload(h, h->pair_allocs, &d.pfan, hp.pfan))) return rc;
    {
        std::vector<int> pet(2 * hp.pelem.size());
        for (size_t i = 0; i < hp.pelem.size(); ++i) {
            pet[2 * i] = hp.pelem[i];
            pet[2 * i + 1] = (int)hp.ptri[4 * i] | ((int)hp.ptri[4 * i + 1] << 10) | ((int)hp.ptri[4 * i + 2] << 20);
        }
        const int *dpet = nullptr;
        if ((rc = dev_upload(h, h->pair_allocs, &dpet, pet))) return rc;
        d.pet = reinterpret_cast<const int2 *>(dpet);
    }
    {
        std::vector<unsigned int> f8;
        nxs_cut::decode_fan8(hp, f8);
        if ((rc = dev_upload(h, h->pair_allocs, &d.pfan8, f8))) return rc;
    }
    d.W2 = m.W2; d.pnbr = nullptr;
    if ((rc = dev_upload(h, h->pair_allocs, &h->pairh.pflags, plan.pflags))) return rc;
    if ((rc = dev_alloc(h, h->pair_allocs, &h->pairh.tickets, 128))) return rc;
    HIPCHK(h, hipMemsetAsync(h->pairh.tickets, 0, 128 * sizeof(unsigned int), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->pairh.nG = plan.nG; h->pairh.nBand = plan.nBand; h->pairh.from_mailbox = 0;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_substep_pair<512, true, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->pair_lds));
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_substep_pair<512, false, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->pair_lds));
    h->pair_ready = true;
    h->pair_depth_built = 2;
    return NXS_OK;
}

int upload_host_patches(nxs_dyn_handle *h, const HostPatches &hp) {
    drop_pool(h, h->patch_allocs);
    release_resident(h);  // (the resident loop's tables describe the patches that go now: the tables, the second exchange buffer, the claim on the device's slots)
    {   // sums over the tables, for nxs_dyn_get_traffic_model
        auto &S1 = h->sums1;
        S1 = nxs_dyn_handle::PatchSums{};
        S1.nP = hp.nP;
        for (int q = 0; q < hp.nP; ++q) {
            S1.M += hp.node_cnt[q]; S1.E += hp.elem_cnt[q]; S1.O += hp.own_cnt[q];
            for (int l = 0; l < hp.elem_cnt[q]; ++l) S1.W += hp.pelem[(size_t)q * hp.Emax + l] >= 0 ? 1. : 0.;
        }
    }
    DevPatches &d = h->dpch;
    d = DevPatches{};
    if (hp.Mmax > 1024) return fail(h, NXS_ERR_INVALID, "a patch stages %d nodes (at most 1024: choose smaller patches)", hp.Mmax);
    d.nP = hp.nP; d.Pmax = hp.Pmax; d.Emax = hp.Emax; d.Mmax = hp.Mmax; d.Wp = hp.Wp;
    int rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.own_cnt, hp.own_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.elem_cnt, hp.elem_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.node_cnt, hp.node_cnt))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pnodes, hp.pnodes))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pelem, hp.pelem))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.ptri, hp.ptri))) return rc;
    if ((rc = dev_upload(h, h->patch_allocs, &d.pfan, hp.pfan))) return rc;
    std::vector<int> pet;
    nxs_cut::pack_pet(hp, pet);
    const int *dpet = nullptr;
    if ((rc = dev_upload(h, h->patch_allocs, &dpet, pet))) return rc;
    d.pet = reinterpret_cast<const int2 *>(dpet);
    // k_prep_fused: the bamg-order rows in patch slots; a table that does not fit this mesh, or patches too large for its LDS, leave
    // the two separate prep kernels in place
    h->prep_lds = 0;
    d.prow = nullptr; d.W1 = h->dm.W1;
    if (!h->h_n2e.empty() && nxs_cut::prep_fused_lds_of(hp) <= 160 * 1024) {   // (several ranks too, since round 5: the rows of the OWN nodes; the ghosts have a pass of their own)
        std::vector<unsigned short> rows;
        if (nxs_cut::build_prep_rows(hp, h->h_n2e.data(), h->dm.W1, h->dm.Nn, rows)) {
            if ((rc = dev_upload(h, h->patch_allocs, &d.prow, rows))) return rc;
            h->prep_lds = nxs_cut::prep_fused_lds_of(hp);
            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_prep_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->prep_lds));
        }
    }
    return NXS_OK;
}

int upload_patches(nxs_dyn_handle *h) {
    drop_pool(h, h->patch_allocs);
    release_resident(h);
    h->dpch = DevPatches{};
    h->fused_lds = 0;
    h->hf_ready = false;
    const DevMesh &m = h->dm;
    nxs_cut::PatchPlan plan;
    const std::string why = nxs_cut::plan_patches(mesh_view(h), h->patch_nodes, h->fused == 4, device_cus(h), plan, NXS_CUT_RES_EPT, !h->no_big_cut, h->band_nodes < 0 ? 48 : h->band_nodes);
    if (!why.empty()) return fail(h, NXS_ERR_INVALID, "%s", why.c_str());
    h->fused_lds = plan.fused_lds;
    h->cut_big = plan.cut_big;
    const HostPatches &hp = plan.hp;
    if (getenv("NXS_DEBUG_PATCHES")) {
        long long se = 0, sm = 0;
        for (int q = 0; q < hp.nP; ++q) { se += hp.elem_cnt[q]; sm += hp.node_cnt[q]; }
        fprintf(stderr, "[nxs] patches: P=%d nP=%d Pmax=%d Emax=%d Mmax=%d Wp=%d avgE=%.1f avgM=%.1f lds=%zu B elems x%.3f%s\n", plan.P, hp.nP, hp.Pmax,
                hp.Emax, hp.Mmax, hp.Wp, (double)se / std::max(hp.nP, 1), (double)sm / std::max(hp.nP, 1), h->fused_lds, (double)se / std::max(m.Ne, 1),
                hp.used_hilbert ? " (cut along a Hilbert curve)" : "");
    }
    h->hp = std::make_shared<HostPatches>(std::move(plan.hp));
    return upload_host_patches(h, *h->hp);
}

// Node-ring patches for the open-water smoother alone (see nxs_cut::plan_smooth_patches).
int build_smooth_patches(nxs_dyn_handle *h, int D) {
    drop_pool(h, h->sm_allocs);
    h->dsm = DevPatches2{};
    h->sm_ready = false;
    const DevMesh &m = h->dm;
    nxs_cut::SmoothPlan plan;
    const std::string why = nxs_cut::plan_smooth_patches(mesh_view(h), h->hp && h->hp->used_hilbert, D, h->h_n2n, h->h_n2n_cnt, m.W2, plan);
    if (!why.empty()) return fail(h, m.No != m.Nn ? NXS_ERR_STATE : NXS_ERR_INVALID, "%s", why.c_str());
    DevPatches2 &d = h->dsm;
    d.nP = plan.nP; d.D = D; d.NDmax = plan.NDmax; d.NSmax = plan.NSmax; d.W2 = m.W2;
    d.own_is_block = plan.own_is_block;
    int rc;
    if ((rc = dev_upload(h, h->sm_allocs, &d.ncnt, plan.ncnt))) return rc;
    if ((rc = dev_upload(h, h->sm_allocs, &d.pnodes, plan.pnodes))) return rc;
    if ((rc = dev_upload(h, h->sm_allocs, &d.pnbr, plan.pnbr))) return rc;
    h->sm_lds = plan.lds;
    h->sm_ready = true;
    if (getenv("NXS_DEBUG_PATCHES")) fprintf(stderr, "[nxs] smoother patches: D=%d nP=%d NDmax=%d NSmax=%d lds=%zu B\n", D, plan.nP, plan.NDmax, plan.NSmax, plan.lds);
    return NXS_OK;
}
