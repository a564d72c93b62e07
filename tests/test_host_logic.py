"""CPU tests of the round-3 host logic (no GPU, no processes): the structured row-stream plan, the partition planner and its
communication model, the generic partitioner's conventions."""
import numpy as np
import pytest


def _expand_plan(po, M):
    """materialise a StreamedCSR from its plan, expanding ("repeat", nrows_block, count, col_shift) items as the library does"""
    ptrs, idxs, vals, base, nblk, nrep, nsent = [np.zeros(1, dtype=np.int64)], [], [], 0, 0, 0, 0
    for it in M.row_plan():
        if it[0] == "block":
            _, row0, B = it
            assert row0 == sum(p.size for p in ptrs) - 1
            ptrs.append(B.ptr[1:] + base); idxs.append(B.idx.astype(np.int64)); vals.append(B.val); base += B.nnz; nblk += 1; nsent += B.shape[0]
        else:
            _, nrb, count, cs = it
            P, I, V = np.concatenate(ptrs), np.concatenate(idxs), np.concatenate(vals)
            nrows = P.size - 1
            p0 = P[nrows - nrb]
            bp, bi, bv = P[nrows - nrb:] - p0, I[p0:], V[p0:]
            for k in range(1, count + 1):
                ptrs.append(bp[1:] + base); idxs.append(bi + k * cs); vals.append(bv); base += bv.size
            nrep += count
    return po.CSR(M.shape, np.concatenate(ptrs), np.concatenate(idxs), np.concatenate(vals)), nblk, nrep, nsent


@pytest.mark.parametrize("nc,order", [((8, 8, 8), 2), ((16, 16, 16), 1), ((12, 8, 10), 2), ((16, 16), 2), ((32, 32, 32), 1)])
def test_row_stream_plan_reproduces_the_operators(po, nc, order):
    """poisson.*_stream().row_plan(): blocks + declared repetitions (gmg_set_operator_rows_repeat) expand to exactly the
    operator `materialize()` gives and the whole-matrix generator gives -- same pointers, columns and value bits; on 3-D meshes
    most planes are declared, not generated, and of the planes that are sent most grid lines."""
    half = tuple(c // 2 for c in nc)
    for M, whole in ((po.poisson_matrix_stream(nc, order), po.poisson_matrix(nc, order)),
                     (po.prolongation_stream(half, order), po.prolongation(half, order)),
                     (po.restriction_stream(half, order), po.prolongation(half, order).transpose())):
        E, nblk, nrep, nsent = _expand_plan(po, M)
        F = M.materialize()
        assert np.array_equal(E.ptr, F.ptr) and np.array_equal(E.idx, F.idx) and np.array_equal(E.val, F.val)
        assert np.array_equal(F.ptr, whole.ptr) and np.array_equal(F.idx, whole.idx) and np.allclose(F.val, whole.val, rtol=0, atol=0)
        if len(nc) == 3 and min(nc) >= 16:
            assert nrep > 0 and nblk <= 12 * 12 and nsent <= 12 * 12 * max(order * c for c in nc), (nblk, nrep, nsent)


def test_partition_planner(pkg):
    """multigpu.plan_partition on BASELINE config 4 (576^3 on 2x2x2 GPUs, 6 levels): the finest level keeps the own | ghost layout,
    levels 1-2 get the overlapping layout with the depth that minimises the modelled pass time, levels whose global size is
    <= 4e5 dofs are replicated (so the 72^3 level is NOT distributed); exchanges per V-cycle on the partitioned levels halve."""
    from gridapsolvers_jl_amd import multigpu as m
    rep_from, depths, table = m.plan_partition(288, 6, 8)
    assert rep_from == 3 and depths[0] == 0 and depths[1] >= 2 and depths[2] >= 5 and depths[3:] == [0, 0, 0]
    per_pass = [row.get("exchanges_per_pass", 0) for row in table[:3]]
    assert per_pass[0] == 10 and per_pass[1] <= 5 and per_pass[2] <= 2
    before = 3 * (2 * 10 + 3)
    after = (2 * 10 + 3) + sum(2 * p + 3 for p in per_pass[1:])
    assert before >= 1.9 * after, (before, after)
    for row in table[1:3]:                                    # the chosen depth is the argmin of the model, and beats the own | ghost schedule
        mp = row["modelled_pass_us"]
        assert min(mp.values()) == mp[str(depths[row["level"]])] < mp["0"]
    # a single GPU plans nothing; tiny problems replicate everything below the finest level
    assert m.plan_partition(128, 4, 1)[1] == [0, 0, 0, 0]
    assert m.plan_partition(32, 3, 2)[0] == 1
    # the model is monotone in the exchange latency: a slower link never chooses a shallower halo
    d_fast = m.plan_partition(288, 6, 8)[1]
    saved = dict(m.MODEL)
    try:
        m.MODEL["exchange_us"], m.MODEL["exchange_overlapped_us"] = 200.0, 220.0
        d_slow = m.plan_partition(288, 6, 8)[1]
    finally:
        m.MODEL.update(saved)
    assert all(a >= b for a, b in zip(d_slow, d_fast))


def test_overlap_geometry_counts(pkg):
    """partition._OverlapGeom: depth-k boxes are clipped at the Dirichlet boundary, every local entry is owned or received exactly
    once, deeper halos only add entries."""
    from gridapsolvers_jl_amd import partition as pa
    prev = None
    for depth in (1, 2, 4, 9):
        g = pa._OverlapGeom((16, 16, 16), (2, 2, 2), 3, 3, depth)
        nbr, sp, si, rp, ri = g.plan()
        assert g.n_own == 7 * 7 * 8 or g.n_own in (7 * 8 * 8, 8 * 8 * 8, 7 * 7 * 7, 7 * 7 * 8, 7 * 8 * 7, 8 * 7 * 7, 8 * 8 * 7, 8 * 7 * 8)
        assert np.array_equal(np.sort(np.concatenate([g.own_idx, ri])), np.arange(g.n_local))
        assert all(g.ext_shape[k] <= 15 for k in range(3))
        if prev is not None:
            assert g.n_local >= prev
        prev = g.n_local


def test_stokes_inputs_by_stencil_replication_equal_the_assembly(pkg):
    """stokes.stokes_system_fast / velocity_hierarchy_fast (what bench.py's config-5 leg feeds the library at sizes the scipy assembly
    needs minutes for) against the assembly they replace: every block, the right-hand side and every level, bit for bit."""
    import importlib
    st = importlib.import_module(pkg.__name__ + ".stokes")
    for n, nlev in ((8, 2), (16, 3)):
        a, b = st.stokes_system(n, 1.0e3), st.stokes_system_fast(n, 1.0e3, with_K=True)
        assert a["sizes"] == b["sizes"] and np.array_equal(a["b"], b["b"])
        assert abs(a["K"] - b["K"]).max() == 0.0 and abs(a["Mp_scaled"].to_scipy() - b["Mp_scaled"].to_scipy()).max() == 0.0
        for i, j in ((0, 0), (0, 1), (1, 0)):
            assert np.array_equal(a["A"][i][j].ptr, b["A"][i][j].ptr) and np.array_equal(a["A"][i][j].idx, b["A"][i][j].idx) \
                and np.array_equal(a["A"][i][j].val, b["A"][i][j].val)
        H1, H2 = st.velocity_hierarchy(n, nlev, 1.0e3), st.velocity_hierarchy_fast(n, nlev, 1.0e3)
        for key in ("mats", "graddiv", "prolongations", "restrictions"):
            for x, y in zip(H1[key], H2[key]):
                assert np.array_equal(x.ptr, y.ptr) and np.array_equal(x.idx, y.idx) and np.array_equal(x.val, y.val), key
        for key in ("star_patches", "interior_patches"):
            for (p1, d1), (p2, d2) in zip(H1[key], H2[key]):
                assert np.array_equal(p1, p2) and np.array_equal(d1, d2)
