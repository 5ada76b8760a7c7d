"""The two fused passes of a rung sub-step (cg_substep_begin / cg_substep_end) against the calls
they stand for (Component.drift, flag_rung_jumps, nullify_Δ | apply_Δmom, convert_Δmom_to_acc,
apply_rung_jumps, set_rungs_N; species.py:2179-2587, main.py:1347-1624): bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('lowest', [0, 2, 4])
@pytest.mark.parametrize('n', [1, 1000, 700001])
def test_substep_passes_equal_the_separate_calls(n, lowest):
    import torch
    from concept_amd.mesh import PotentialMesh
    L, N_rungs = 37.0, 8
    mesh = PotentialMesh(16, L)
    g = torch.Generator(device='cuda').manual_seed(n + lowest)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=g)*L*(1 - 1e-13)
    mom = torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=g)
    acc = torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=g) \
        * torch.exp(4*torch.randn((n, 1), dtype=torch.float64, device='cuda', generator=g))
    acc[::7] = 0
    rung = torch.randint(0, 6, (n,), device='cuda', generator=g).to(torch.int8)
    rng = np.random.default_rng(5)
    integrals = rng.uniform(0.1, 1.0, 3*N_rungs - 1)
    integrals[3] = 0.0                       # a rung whose kick is empty
    integrals[N_rungs:2*N_rungs:2] = -1      # no downward jump this sub-step
    conv = rng.uniform(0.5, 2.0, 3*N_rungs - 1)
    rf_up, rf_down, dtm = 1.3, 2.1, 0.37
    up = lambda a: torch.as_tensor(a, device='cuda')   # noqa: E731
    # the separate calls
    p0, m0, d0, r0, j0 = pos.clone(), mom.clone(), acc.clone(), rung.clone(), rung.clone()
    mesh.drift(p0, m0, dtm)
    flagged0 = mesh.flag_rung_jumps(d0, r0, j0, lowest, up(integrals), rf_up, rf_down, N_rungs)
    mesh.dmom_nullify(d0, r0, lowest)
    # the pass
    p1, m1, d1, r1, j1 = pos.clone(), mom.clone(), acc.clone(), rung.clone(), rung.clone()
    any_out = torch.zeros(1, dtype=torch.int32, device='cuda')
    after = torch.full((N_rungs,), -1, dtype=torch.int64, device='cuda')
    mesh.substep_begin(p1, m1, d1, r1, j1, dtm, True, lowest, integrals, rf_up, rf_down, N_rungs,
                       any_out, after)
    assert torch.equal(p0, p1) and torch.equal(d0, d1) and torch.equal(j0, j1)
    assert bool(any_out.item()) == flagged0
    assert n < 1000 or (flagged0 and bool((j1 >= N_rungs).any()) and bool((j1 >= 2*N_rungs).any()))
    # drift alone, flag alone
    p2 = pos.clone()
    mesh.substep_begin(p2, m1, None, r1, j1.clone(), dtm, False, lowest, None, 0.0, 0.0, N_rungs,
                       any_out)
    assert torch.equal(p2, p0)
    d3, j3 = acc.clone(), rung.clone()
    mesh.substep_begin(pos.clone(), m1, d3, rung, j3, None, True, lowest, integrals, rf_up,
                       rf_down, N_rungs, any_out)
    assert torch.equal(d3, d0) and torch.equal(j3, j0)
    # a kick arrives in Δmom
    kick = torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=g)
    d0 += kick
    d1 += kick
    mesh.dmom_apply(m0, d0, r0, lowest)
    mesh.dmom_to_acc(d0, r0, j0, lowest, up(conv), flagged0)
    mesh.apply_rung_jumps(r0, j0, N_rungs)
    counts0 = mesh.rung_populations(r0, N_rungs)
    counts1 = torch.zeros(N_rungs, dtype=torch.int64, device='cuda')
    mesh.substep_end(m1, d1, r1, j1, True, lowest, conv, N_rungs, counts1)
    assert torch.equal(m0, m1) and torch.equal(d0, d1)
    assert torch.equal(r0, r1) and torch.equal(j0, j1) and torch.equal(r1, j1)
    assert torch.equal(counts0, counts1) and int(counts1.sum()) == n
    # ... which the first pass had counted before the jumps were applied
    assert torch.equal(after, counts0)
    mesh.substep_end(m1.clone(), d1.clone(), r1.clone(), j1.clone(), True, lowest, conv, N_rungs,
                     None)
    # a component that received nothing: the jumps and the populations only
    r4, j4 = rung.clone(), j3.clone()
    m4, d4 = mom.clone(), acc.clone()
    mesh.substep_end(m4, d4, r4, j4, False, lowest, None, N_rungs, counts1)
    assert torch.equal(m4, mom) and torch.equal(d4, acc) and torch.equal(r4, r1)
    assert torch.equal(counts0, counts1)
    mesh.close()


@pytest.mark.parametrize('lowest', [0, 3])
def test_deferred_pass_runs_with_the_cell_list(lowest):
    """substep_begin(defer=True): nothing happens until the next shortrange_cells() on the same
    positions, whose counting pass drifts, flags and nullifies every particle as it bins it — the
    arrays and the list equal those of the pass followed by the list; a pass still pending is
    run by any other call that touches particles."""
    import torch
    from concept_amd import commons
    from concept_amd.mesh import PotentialMesh
    N, L, N_rungs, n = 64, 64.0, 8, 50021
    mesh = PotentialMesh(N, L)
    g = torch.Generator(device='cuda').manual_seed(77 + lowest)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=g)*L*(1 - 1e-13)
    mom = torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=g)
    acc = torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=g) \
        * torch.exp(4*torch.randn((n, 1), dtype=torch.float64, device='cuda', generator=g))
    rung = torch.randint(0, 6, (n,), device='cuda', generator=g).to(torch.int8)
    integrals = np.random.default_rng(3).uniform(0.1, 1.0, 3*N_rungs - 1)
    rf_up, rf_down, dtm = 1.3, 2.1, 0.41
    nt = int(L/(4.5*1.25*L/N)*(1 + commons.machine_ϵ))
    any0 = torch.zeros(1, dtype=torch.int32, device='cuda')
    any1 = torch.zeros(1, dtype=torch.int32, device='cuda')
    p0, d0, j0 = pos.clone(), acc.clone(), rung.clone()
    mesh.substep_begin(p0, mom, d0, rung, j0, dtm, True, lowest, integrals, rf_up, rf_down,
                       N_rungs, any0)
    list0 = mesh.shortrange_cells(p0, nt, L/nt, (rung, j0, lowest))
    p1, d1, j1 = pos.clone(), acc.clone(), rung.clone()
    after0 = torch.zeros(N_rungs, dtype=torch.int64, device='cuda')
    after1 = torch.zeros(N_rungs, dtype=torch.int64, device='cuda')
    mesh.substep_begin(pos.clone(), mom, acc.clone(), rung, rung.clone(), dtm, True, lowest,
                       integrals, rf_up, rf_down, N_rungs, any0, after0)
    mesh.substep_begin(p1, mom, d1, rung, j1, dtm, True, lowest, integrals, rf_up, rf_down,
                       N_rungs, any1, after1, defer=True)
    torch.cuda.synchronize()
    assert torch.equal(p1, pos) and torch.equal(d1, acc)      # nothing yet
    list1 = mesh.shortrange_cells(p1, nt, L/nt, (rung, j1, lowest))
    assert torch.equal(p1, p0) and torch.equal(d1, d0) and torch.equal(j1, j0)
    assert torch.equal(after0, after1) and int(after1.sum()) == n
    rr, jj = rung.clone(), j1.clone()
    mesh.apply_rung_jumps(rr, jj, N_rungs)
    assert torch.equal(after1, mesh.rung_populations(rr, N_rungs))
    assert int(any0.item()) == int(any1.item()) == 1
    assert torch.equal(list0[1], list1[1])                     # the cells' offsets
    if lowest:
        assert torch.equal(list0[3], list1[3])                 # active rows per cell
    # the same members in every cell (the order inside a cell is the atomics')
    cell = torch.repeat_interleave(torch.arange(8*nt**3, device='cuda'),
                                   (list0[1][1:] - list0[1][:-1]).long())
    for lst in (list0, list1):
        assert torch.equal(lst[2][:n], p0[lst[0][:n].long()])
    key0 = torch.sort(cell*n + list0[0][:n].long()).values
    key1 = torch.sort(cell*n + list1[0][:n].long()).values
    assert torch.equal(key0, key1)
    # a pending pass and another list (other positions): the pass runs by itself first
    p2, d2, j2 = pos.clone(), acc.clone(), rung.clone()
    mesh.substep_begin(p2, mom, d2, rung, j2, dtm, True, lowest, integrals, rf_up, rf_down,
                       N_rungs, any1, after1, defer=True)
    other = mesh.shortrange_cells(pos, nt, L/nt)
    assert torch.equal(p2, p0) and torch.equal(d2, d0) and torch.equal(j2, j0)
    assert torch.equal(other[2][:n], pos[other[0][:n].long()])
    # ... or by a flush
    p3, d3, j3 = pos.clone(), acc.clone(), rung.clone()
    mesh.substep_begin(p3, mom, d3, rung, j3, dtm, True, lowest, integrals, rf_up, rf_down,
                       N_rungs, any1, None, defer=True)
    mesh.substep_flush()
    assert torch.equal(p3, p0) and torch.equal(d3, d0) and torch.equal(j3, j0)
    mesh.close()


def test_permute_rows_carries_every_column_in_one_pass():
    """cg_permute_rows: dst[k][q] = src[k][perm[q]] for a Component's other columns (Δmom,
    identifiers, the order column, the rung arrays) — against torch.index_select per column;
    more columns than one launch takes; what the store does with them after a tile sort."""
    import torch
    from concept_amd.lib import ConceptGPUError
    from concept_amd.mesh import PotentialMesh
    mesh = PotentialMesh(16, 16.0)
    g = torch.Generator(device='cuda').manual_seed(9)
    n, cap = 100003, 100100
    perm = torch.randperm(n, device='cuda', generator=g)
    cols = [torch.randn((cap, 3), dtype=torch.float64, device='cuda', generator=g),
            torch.randint(0, 2**40, (cap,), device='cuda', generator=g),
            torch.randint(0, 2**40, (cap,), device='cuda', generator=g),
            torch.randint(-8, 8, (cap,), device='cuda', generator=g).to(torch.int8),
            torch.randint(-8, 8, (cap,), device='cuda', generator=g).to(torch.int8),
            torch.randn((cap, 5), dtype=torch.float32, device='cuda', generator=g),
            torch.randint(0, 2**30, (cap,), device='cuda', generator=g).to(torch.int32),
            torch.randn((cap, 2), dtype=torch.float64, device='cuda', generator=g),
            torch.randn((cap, 7), dtype=torch.float64, device='cuda', generator=g),
            torch.randint(0, 255, (cap, 3), device='cuda', generator=g).to(torch.uint8)]
    outs = [torch.zeros_like(c) for c in cols]
    mesh.permute_rows(perm, list(zip(cols, outs)))
    for c, o in zip(cols, outs):
        assert torch.equal(o[:n], torch.index_select(c[:n], 0, perm))
        assert bool((o[n:] == 0).all())
    mesh.permute_rows(perm[:0], list(zip(cols, outs)))       # nothing to do
    with pytest.raises(ConceptGPUError, match='permute_rows'):
        mesh.permute_rows(perm, [(cols[0], outs[1])])
    mesh.close()
