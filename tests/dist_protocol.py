"""CPU restatement of the multi-GPU CG protocol of magnetite_amd/csrc/api.hip (launch_block, cg_phase), run with
world_size ranks over gloo.  Each rank owns a contiguous range of nodes of a spatial ordering and applies only ITS
rows of K_ff (oracle CSR); per iteration exactly the library's two collectives are issued:
    all_reduce([p.q partial])                       after the operator
    all_reduce([r.r partial | r on interface nodes]) after the update
ghost p is advanced locally from the exchanged r (p = -r + beta p), and the solution is assembled by a final
all-reduce.  Checks the result against the single-process oracle CG.

`--fused` restates the default one-launch iteration instead (cg_phase_fused / k_cg_fused<COMM>): ONE in-place
all-reduce per iteration on a parity-double-buffered exchange buffer
    [r.r | p.q | r.q | q.q partials, G slots each, summed slot by slot | q on interface nodes (owner's value, 0 elsewhere)]
launch j reads the all-reduced buffer [j & 1] (sums and ghost q of iterate j-1), advances owned and ghost records,
and fills buffer [(j & 1) ^ 1]; beta's numerator is expanded from the exact sums of the previous iterate.

`--onchip` restates the multi-GPU on-chip CG (persist.hip, k_cg_persist<.., MG>): no collective per iteration -- every
rank keeps r, p of its halo nodes itself, the owner of an interface node delivers its q into the INBOX of exactly the
ranks that read it, every rank delivers its four sums into every rank's inbox and adds the R records in rank order
(inboxes are modelled by an all-gather of what each rank would store)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist

    import oracle
    from magnetite_amd import meshgen

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    p = meshgen.config_fixed_left_pull_right(meshgen.shuffle(meshgen.plate_with_holes(20, 30, 1.0, 1.5), 5))
    N = p.mesh.num_nodes
    K = oracle.assemble_sparse(p.xy_flat, p.conn_flat, p.poisson_ratio, p.youngs_modulus, p.part_thickness)
    Kf = sp.csr_matrix((K.val, K.col, K.rowptr), shape=(K.n, K.n))
    free = (p.u_known == 0)
    M = sp.diags(free.astype(np.float64))
    A = (M @ Kf @ M).tocsr()                      # K_ff embedded in full length, as the GPU operator applies it
    b = np.where(free, p.f_in - Kf @ np.where(free, 0.0, p.u_in), 0.0)

    # ordering + contiguous ranges (the library uses Hilbert order and tile granularity; any spatial order works here)
    B = 16
    order = np.lexsort((p.mesh.xy[:, 0], np.floor(p.mesh.xy[:, 1] * 6)))
    T = (N + B - 1) // B
    lo = [min((T * s // world) * B, N) for s in range(world + 1)]
    owner_of_pos = np.searchsorted(lo, np.arange(N), side="right") - 1
    owner = np.empty(N, dtype=np.int64)
    owner[order] = owner_of_pos
    dof_owner = np.repeat(owner, 2)
    mine = dof_owner == rank
    # interface: nodes read by a rank (columns of its rows) that it does not own -- union over ranks, sorted
    node_adj = sp.csr_matrix((np.ones(K.nnz), K.col // 2, K.rowptr), shape=(K.n, N))
    iface = set()
    for s in range(world):
        rows = np.where(dof_owner == s)[0]
        cols = np.unique(node_adj[rows].indices)
        iface.update(int(c) for c in cols if owner[c] != s)
    iface = np.array(sorted(iface), dtype=np.int64)
    iface_dofs = np.stack([2 * iface, 2 * iface + 1], axis=1).reshape(-1)
    iface_mine = dof_owner[iface_dofs] == rank
    Arows = A[np.where(mine)[0]]

    def allreduce(v):
        t = torch.from_numpy(v)
        dist.all_reduce(t)
        return v

    if "--onchip" in sys.argv:
        readers = {}                                   # interface DOF -> ranks that read it (rows of theirs touch it)
        for s_ in range(world):
            rows = np.where(dof_owner == s_)[0]
            cols = np.unique(A[rows].indices)
            for c in cols:
                if dof_owner[c] != s_:
                    readers.setdefault(int(c), set()).add(s_)
        it, x = onchip_protocol(K.n, b, Arows, mine, dof_owner, readers, rank, world)
        finish(p, free, x, mine, it, iface, rank, world, allreduce)
        return

    if "--fused" in sys.argv:
        it, x = fused_protocol(K.n, b, Arows, mine, iface_dofs, iface_mine, rank, allreduce)
        finish(p, free, x, mine, it, iface, rank, world, allreduce)
        return

    x = np.zeros(K.n)
    r = -b.copy()                                  # every rank starts from the full right-hand side
    pp = np.zeros(K.n)
    rr = allreduce(np.array([np.dot(r[mine], r[mine])]))[0]
    rr_prev, it, target = rr, 0, 1e-4
    while it < 100000:
        if it >= 1 and np.sqrt(rr) <= target:
            break
        beta = rr / rr_prev if it > 0 else 1.0
        pn = np.zeros(K.n)
        pn[mine] = -r[mine] + beta * pp[mine]
        notmine_if = iface_dofs[~iface_mine]
        pn[notmine_if] = -r[notmine_if] + beta * pp[notmine_if]   # ghost p from exchanged r
        q = Arows @ pn
        pq = allreduce(np.array([np.dot(pn[mine], q)]))[0]
        alpha = rr / pq
        x[mine] += alpha * pn[mine]
        r[mine] += alpha * q
        buf = np.zeros(1 + iface_dofs.size)
        buf[0] = np.dot(r[mine], r[mine])
        buf[1:][iface_mine] = r[iface_dofs[iface_mine]]
        allreduce(buf)
        r[notmine_if] = buf[1:][~iface_mine]
        rr_prev, rr, pp = rr, buf[0], pn
        it += 1
    finish(p, free, x, mine, it, iface, rank, world, allreduce)


def finish(p, free, x, mine, it, iface, rank, world, allreduce):
    import torch.distributed as dist

    import oracle
    x[~mine] = 0.0
    allreduce(x)
    u = np.where(free, x, p.u_in)
    ref = oracle.run(p.xy_flat, p.conn_flat, p.u_known, p.u_in, p.f_in, p.youngs_modulus, p.poisson_ratio,
                     p.part_thickness, path="sparse")
    err = np.linalg.norm(u - ref["u"]) / np.linalg.norm(ref["u"])
    print(f"rank {rank}/{world}: iface nodes {iface.size}, iterations {it} (oracle {ref['iterations']}), rel-L2 {err:.2e}",
          flush=True)
    ok = err <= 1e-8 and abs(it - ref["iterations"]) <= max(3, ref["iterations"] // 50) and iface.size > 0
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def onchip_protocol(n, b, Arows, mine, dof_owner, readers, rank, world, target=1e-4):
    """One rank of the multi-GPU on-chip CG.  State: r, q, p, x of owned DOFs; r, p of the halo DOFs it reads."""
    import torch.distributed as dist

    halo = np.array(sorted(c for c, rs in readers.items() if rank in rs), dtype=np.int64)  # what this rank reads
    keep = mine.copy()
    keep[halo] = True                              # DOFs this rank carries at all
    r, q, pp, x = np.where(keep, -b, 0.0), np.zeros(n), np.zeros(n), np.zeros(n)

    def exchange(record, qvals):
        """Inboxes: every rank ends up with all R records and with the q of the halo DOFs it reads."""
        deliveries = {}
        for c, rs in readers.items():
            if dof_owner[c] == rank:
                for dst in rs:                     # the owner stores into each reader's inbox, nobody else's
                    deliveries.setdefault(dst, {})[c] = qvals[c]
        box = [None] * world
        dist.all_gather_object(box, (record, deliveries))
        total = np.zeros(4)
        for s_ in range(world):                    # rank order: the same bits on every rank
            total = total + box[s_][0]
        got = {}
        for s_ in range(world):
            got.update(box[s_][1].get(rank, {}))
        assert set(got) == set(halo.tolist())
        return total, got

    rec = np.array([np.dot(b[mine], b[mine]), 1.0 if rank == 0 else 0.0, 0.0, 0.0])
    S, got = exchange(rec, q)
    j = 0
    while j < 200000:
        rr = S[0]
        it_done = j - 1
        if it_done >= 1 and np.sqrt(rr) <= target:
            return it_done, x
        alpha = rr / S[1]
        beta = (rr + 2.0 * alpha * S[2] + alpha * alpha * S[3]) / rr
        for c, v in got.items():
            q[c] = v                               # q_{j-1} of the halo from the inbox
        x[mine] += alpha * pp[mine]
        r[keep] += alpha * q[keep]
        pn = np.zeros(n)
        pn[keep] = -r[keep] + beta * pp[keep]      # owned and halo alike, by the same recurrences
        qn = Arows @ pn
        q[mine] = qn
        pp = pn
        rec = np.array([np.dot(r[mine], r[mine]), np.dot(pn[mine], qn), np.dot(r[mine], qn), np.dot(qn, qn)])
        S, got = exchange(rec, q)
        j += 1
    return j, x


def fused_protocol(n, b, Arows, mine, iface_dofs, iface_mine, rank, allreduce, target=1e-4, G=4):
    """Records {r, q, p} of iterate j-1 -> iterate j in one step per iteration, as k_cg_fused<COMM> does."""
    own_if = iface_dofs[iface_mine]                # interface DOFs this rank owns: their q goes into the buffer
    ghost_if = iface_dofs[~iface_mine]             # ... owned elsewhere: their q comes out of it
    nq = iface_dofs.size
    cbuf = [np.zeros(4 * G + nq), np.zeros(4 * G + nq)]
    grid = 1 + rank % G                            # ranks may run different grids: unused slots are written as zeros
    r, q, pp, x = -b.copy(), np.zeros(n), np.zeros(n), np.zeros(n)

    def partials(v):                               # `grid` partial sums in slots [0, grid), zeros above
        out = np.zeros(G)
        for k, chunk in enumerate(np.array_split(v, grid)):
            out[k] = chunk.sum()
        return out

    c0 = cbuf[0]
    c0[0:G] = partials(b[mine] * b[mine])          # fused_init: b.b partials, "p.q" = 1 on slot 0
    c0[G] = 1.0
    allreduce(c0)
    j = 0
    while j < 200000:
        cin, cout = cbuf[j & 1], cbuf[(j & 1) ^ 1]
        S = [cin[c * G:(c + 1) * G].sum() for c in range(4)]
        rr = S[0]
        it_done = j - 1
        if it_done >= 1 and np.sqrt(rr) <= target:
            return it_done, x
        alpha = rr / S[1]
        beta = (rr + 2.0 * alpha * S[2] + alpha * alpha * S[3]) / rr
        q[ghost_if] = cin[4 * G:][~iface_mine]     # ghost q of iterate j-1 straight from the exchange buffer
        upd = np.concatenate([np.where(mine)[0], ghost_if])
        x[mine] += alpha * pp[mine]
        r[upd] += alpha * q[upd]
        pn = np.zeros(n)
        pn[upd] = -r[upd] + beta * pp[upd]
        qn = Arows @ pn                            # owned rows only
        q[mine] = qn
        pp = pn
        cout[0 * G:1 * G] = partials(r[mine] * r[mine])
        cout[1 * G:2 * G] = partials(pn[mine] * qn)
        cout[2 * G:3 * G] = partials(r[mine] * qn)
        cout[3 * G:4 * G] = partials(qn * qn)
        qb = cout[4 * G:]
        qb[:] = 0.0                                # this rank's zero for interface nodes it does not own
        qb[iface_mine] = q[own_if]
        allreduce(cout)
        j += 1
    return j, x


if __name__ == "__main__":
    main()
