"""Worker of tests/test_krylov.py: one rank of a row-distributed CG / BiCGStab / SpMV (ranks may share GPU 0).
argv: out_dir, mode ("numpy": host logic only, no GPU; "host": GPU + the caller's communicator (gloo);
"rccl": GPU + RCCL)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch, torch.distributed as dist
import cases
from nextsim_amd import krylov, mesh as M

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); out = sys.argv[1]; mode = sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=world)
report = {"rank": rank, "ok": False}
try:
    gm = cases.global_mesh("small")
    A, b, xs = cases.mesh_operator(gm, nonsymmetric=False)
    N, bN, _ = cases.mesh_operator(gm, nonsymmetric=True)
    lm = M.localize(gm, world)[rank]
    n_own = int(lm.local_ndof)
    own = lm.node_gid[:n_own]
    ts, tr = int(lm.send_offsets[-1]), int(lm.recv_offsets[-1])

    def exchange(send, recv):
        reqs, bufs = [], []
        for k, q in enumerate(lm.send_procs):
            a, e = int(lm.send_offsets[k]), int(lm.send_offsets[k + 1])
            reqs.append(dist.isend(torch.from_numpy(np.array(send[a:e], copy=True)), int(q)))
        for k, q in enumerate(lm.recv_procs):
            a, e = int(lm.recv_offsets[k]), int(lm.recv_offsets[k + 1])
            t = torch.empty(e - a, dtype=torch.float64)
            reqs.append(dist.irecv(t, int(q))); bufs.append((a, e, t))
        for r_ in reqs:
            r_.wait()
        for a, e, t in bufs:
            recv[a:e] = t.numpy()

    def allreduce(vals):
        t = torch.from_numpy(np.array(vals, copy=True))
        dist.all_reduce(t)
        vals[:] = t.numpy()

    rp, ci, va, n_loc = krylov.localize_system(A.indptr, A.indices, A.data, lm)
    x_glob = np.random.default_rng(5).normal(size=gm.num_nodes)
    want = (A @ x_glob)[own]
    if mode == "numpy":
        # the partition's rows + halo lists reproduce the global product (what the device kernels are given)
        import scipy.sparse as sp
        loc = sp.csr_matrix((va, ci, rp), shape=(n_own, n_loc))
        v = np.zeros(n_loc); v[:n_own] = x_glob[own]
        send = v[lm.send_index].copy(); recv = np.zeros(tr)
        exchange(send, recv)
        v[lm.recv_index] = recv
        assert np.array_equal(v, x_glob[lm.node_gid])
        got = loc @ v
        report["spmv_err"] = float(np.abs(got - want).max() / np.abs(want).max())
        d = np.array([float(v[:n_own] @ v[:n_own])]); allreduce(d)
        report["dot_err"] = float(abs(d[0] - x_glob @ x_glob) / (x_glob @ x_glob))
    else:
        s = krylov.Solver(device=int(os.environ.get("NXS_TEST_DEVICE", "0")))
        s.set_matrix(rp, ci, va, n_cols=n_loc)
        s.set_halo(lm)
        if mode == "rccl":
            from nextsim_amd import dynamics
            ids = [dynamics.FiniteElementDynamics.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                s.comm_init(ids[0], rank, world)
            except dynamics.NxsError as e:
                report["comm_error"] = str(e)
                raise
        else:
            s.set_comm_fns(exchange, allreduce, ts, tr)
        got, _ = s.spmv(x_glob[own])
        report["spmv_err"] = float(np.abs(got - want).max() / np.abs(want).max())
        x, info = s.solve(b[own], method=krylov.CG, rtol=1e-12, max_iter=5000)
        report["cg"] = dict(info, err=float(np.abs(x - xs[own]).max() / np.abs(xs).max()))
        rp2, ci2, va2, _ = krylov.localize_system(N.indptr, N.indices, N.data, lm)
        s.set_matrix(rp2, ci2, va2, n_cols=n_loc)
        s.set_halo(lm)
        x, info = s.solve(bN[own], method=krylov.BICGSTAB, rtol=1e-12, max_iter=5000)
        report["bicgstab"] = dict(info, err=float(np.abs(x - xs[own]).max() / np.abs(xs).max()))
        report["comm_stats"] = s.comm_stats()
        if mode == "rccl" and world == 1:   # a sum over one rank: the bits of the solve without any communicator
            s1 = krylov.Solver(device=int(os.environ.get("NXS_TEST_DEVICE", "0")))
            s1.set_matrix(rp2, ci2, va2, n_cols=n_loc)
            x1, info1 = s1.solve(bN[own], method=krylov.BICGSTAB, rtol=1e-12, max_iter=5000)
            report["same_bits_as_no_communicator"] = bool(np.array_equal(x, x1) and info1["iterations"] == info["iterations"])
            s1.close()
        s.close()
    report["ok"] = True
except Exception as e:  # noqa: BLE001
    import traceback
    report["error"] = repr(e) + traceback.format_exc()
json.dump(report, open(os.path.join(out, f"kr{rank}.json"), "w"))
try:
    dist.barrier(); dist.destroy_process_group()
except Exception:
    pass
