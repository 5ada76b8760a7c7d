"""The kernels bench.py TIMES, at the sizes it times them, against the C oracle
(VERDICT r2 "next round" item 1).

bench.py's fused step is
    cg_deposit_cic_tiled (first step) / cg_deposit_cic_regions -> cg_poisson_solve ->
    cg_predict_regions -> cg_gather_kick_drift_scatter
entered after one cg_drift_sort.  (a) runs exactly that sequence for two consecutive steps (the
second reads regions with gaps) at BASELINE configs[0] (128^3 / 256^3) and configs[1]
(256^3 / 512^3) against oracle.pm_long_range + oracle.drift; (b) runs it at the metric's own
size (2^28 particles / 1024^3 mesh) against the separate passes on a strided sample plus
size-independent properties.

Bars (reference: test/pure_python_pm/analyze.py:125, 1e-10 on positions over a whole run):
  * CIC indices of a step whose input positions are bit-identical: bit-exact;
  * the drift of the fused pass: bit-exact against the oracle's drift applied to the pass's own
    inputs (pos before, momenta after the kick) — positions of the whole trajectory therefore
    differ from the oracle's only through the kick's rounding: <= 1e-13 L;
  * momenta: <= 1e-12 of the rms kick (+ the rounding floor of the momenta the kick is added
    to);
  * every particle inside the region of its (tile, bucket); cg_error_flags == 0.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch


def _live(torch, start, count, cap):
    st, ct = start.long(), count.long()
    slot = torch.arange(cap, device=start.device)
    k = (torch.searchsorted(st, slot, right=True) - 1).clamp(max=ct.numel() - 1)
    return (slot - st[k]) < ct[k], k


def _keys(torch, mesh, pos, N, g=2):
    """8*tile + bucket of positions, from the library's own CIC index map (cg_tiles.h)"""
    t_ext, nt = mesh.tile_extent, mesh.tiles_per_dim
    cells = torch.remainder(mesh.cic_indices(pos, False) - g, N)
    tile = ((cells[:, 0]//t_ext)*nt + cells[:, 1]//t_ext)*nt + cells[:, 2]//t_ext
    last = (cells % t_ext) == (t_ext - 1)
    return tile*8 + last[:, 0]*4 + last[:, 1]*2 + last[:, 2]*1


def _clustered(rng, n, L, nclumps=8, frac=0.8, sigma=1/10):
    """bench.py --dist clustered in small: Gaussian clumps over a uniform background (as wide
    in cells as the bench's: regions sized for the present populations + 25 % hold the next)"""
    pos = rng.uniform(0, L, (n, 3))
    centres = rng.uniform(0, L, (nclumps, 3))
    in_clump = rng.uniform(size=n) < frac
    blob = centres[rng.integers(0, nclumps, n)] + rng.normal(0, sigma*L, (n, 3))
    pos[in_clump] = np.mod(blob[in_clump], L)
    return np.minimum(pos, np.nextafter(L, 0))


@pytest.mark.parametrize('npart,N,dist', [(128, 256, 'uniform'), (256, 512, 'uniform'),
                                          (128, 256, 'clustered')])
def test_bench_sequence_two_steps_vs_oracle(npart, N, dist):
    """(the clustered box: 4096 tiles of 512 particles on average, several thousand in the
    clumps' tiles — the tile kernels run in the order of cgk_tile_order, heavy tiles first)"""
    import torch
    from concept_amd.mesh import PotentialMesh
    from oracle import oracle
    n, L = npart**3, 400.0
    mass, G, dt, order = 1.7, 0.9, 0.03, 2
    dtm, kick_factor = dt/mass, mass*(-dt)
    rng = np.random.default_rng(1000 + npart)
    pos0 = rng.uniform(0, L, (n, 3)) if dist == 'uniform' else _clustered(rng, n, L)
    # ~0.3 cells per drift: particles change bucket and tile, and wrap around the box
    mom0 = rng.normal(0, 0.3*(L/N)*mass/dt/3**0.5, (n, 3))
    mesh = PotentialMesh(N, L, nghosts=2)
    cap = mesh.region_capacity(n)
    dev = 'cuda'

    def grown(a, dtype, width):
        t = torch.full((cap,) + ((width,) if width else ()), -7, dtype=dtype, device=dev)
        if a is not None:
            t[:n] = torch.as_tensor(a, device=dev)
        return t
    pos, mom = grown(pos0, torch.float64, 3), grown(mom0, torch.float64, 3)
    ids = grown(np.arange(n), torch.int64, 0)
    pos2, mom2, ids2 = grown(None, torch.float64, 3), grown(None, torch.float64, 3), \
        grown(None, torch.int64, 0)
    table = mesh.new_tile_table()
    # bench.py: one drift + sort into tile order up front
    mesh.drift_sort(pos[:n], mom[:n], ids[:n], pos2[:n], mom2[:n], ids2[:n], dtm, table)
    pos, pos2, mom, mom2, ids, ids2 = pos2, pos, mom2, mom, ids2, ids
    pos_o = oracle.drift(pos0.copy(), mom0, dtm, L)
    mom_o = mom0.copy()
    perm = ids[:n].cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(pos[:n].cpu().numpy(), pos_o[perm])   # A11 bit-exact
    start, count = table[:mesh.table_entries], None
    tabs = [mesh.new_region_table(), mesh.new_region_table()]
    contribution = oracle.deposit_contribution(mass, dt, dt, N, L)
    C, _ = oracle.poisson_constants(L, G, None)
    floor = 2.3e-16*np.abs(mom0).max()
    for step in range(2):
        # ---- oracle: one long-range kick at the present positions, then the drift
        mom_before_o = mom_o.copy()
        out = oracle.pm_long_range(pos_o, mom_o, mass=mass, boxsize=L, gridsize=N, G_Newton=G,
                                   dt_1=dt, dt_dens=dt, dt_kick=dt, diff_order=order,
                                   want_indices=(step == 0))
        # ---- the bench's sequence
        if count is None:
            live_in = torch.zeros(cap, dtype=torch.bool, device=dev)
            live_in[:n] = True
            mesh.deposit_tiled(pos[:n], table, contribution, accumulate=False)
        else:
            live_in, _ = _live(torch, start, count, cap)
            mesh.deposit_regions(pos, start, count, contribution)
        pin, idin = pos[live_in], ids[live_in].cpu().numpy()
        if step == 0:
            # A1: indices bit-exact (positions are bit-identical to the oracle's here)
            sample = slice(0, n, 5)
            idx = mesh.cic_indices(pin[sample].contiguous(), False).cpu().numpy()
            assert np.array_equal(idx, out['cic_index_deposit'][idin[sample]])
            idx = mesh.cic_indices(pin[sample].contiguous(), True).cpu().numpy()
            assert np.array_equal(idx, out['cic_index_gather'][idin[sample]])
            # A1/A2: the density mesh against the oracle's folded deposit
            g = 2
            dens_o = np.zeros((N, N, N))
            ix = (np.arange(N + 2*g) - g) % N
            np.add.at(dens_o, (ix[:, None, None], ix[None, :, None], ix[None, None, :]),
                      out['grid_deposit'])
            dens = mesh.fetch_real()[:, :, :N]
            assert np.abs(dens - dens_o).max() <= 1e-12*np.sqrt((dens_o**2).mean())
            del dens, dens_o
        else:
            # the region deposit of the gapped layout: total mass
            tot = float(torch.as_tensor(mesh.fetch_real()[:, :, :N]).sum())
            assert abs(tot - n*contribution) <= 1e-9*n*abs(contribution)
        mesh.poisson_solve(4, C, False, 0.0)
        if step == 0:
            phi = mesh.fetch_real()[:, :, :N]
            phi_o = out['grid_potential'][2:-2, 2:-2, 2:-2]
            assert np.abs(phi - phi_o).max() <= 1e-12*np.sqrt((phi_o**2).mean())
            del phi, phi_o
        del out
        start_out, count_out = tabs[step]
        mesh.predict_regions(start, count, start_out)
        mesh.gather_kick_drift_scatter(pos, mom, ids, start, count, pos2, mom2, ids2, start_out,
                                       count_out, order, kick_factor, dtm)
        assert mesh.error_flags() == 0
        heavy = mesh.tile_order()
        if dist == 'clustered':
            # the pass ran the heavy tiles of these populations first, by falling (class of)
            # population: units of a third of the threshold, 63 at most
            if count is None:
                tab = start.cpu().numpy().astype(np.int64)
                pops = tab[8::8] - tab[:-8:8]
            else:
                pops = count.cpu().numpy().astype(np.int64).reshape(-1, 8).sum(1)
            thr = max(1536, 3*n//(2*mesh.ntiles))
            want = np.flatnonzero(pops > thr)
            assert 8 < want.size <= mesh.ntiles//8
            assert np.array_equal(np.sort(heavy), want)
            assert (np.diff(np.minimum(pops[heavy]//(thr//3), 63)) <= 0).all()
        else:
            assert heavy is not None and heavy.size == 0
        pos_prev = np.empty((n, 3))
        pos_prev[idin] = pin.cpu().numpy()
        del pin
        pos, pos2, mom, mom2, ids, ids2 = pos2, pos, mom2, mom, ids2, ids
        start, count = start_out, count_out
        # ---- compare
        live, region = _live(torch, start, count, cap)
        assert int(live.sum()) == n and int(count.long().sum()) == n
        st = start.long()
        assert bool((st[1:] - st[:-1] >= count.long()).all()) and int(st[-1]) <= cap
        got_i = ids[live].cpu().numpy()
        assert np.array_equal(np.sort(got_i), np.arange(n))
        # every particle inside the region of its (tile, bucket)
        assert torch.equal(_keys(torch, mesh, pos[live].contiguous(), N), region[live])
        m_gpu, p_gpu = np.empty((n, 3)), np.empty((n, 3))
        m_gpu[got_i] = mom[live].cpu().numpy()
        p_gpu[got_i] = pos[live].cpu().numpy()
        kick_o = mom_o - mom_before_o
        scale = np.sqrt((kick_o**2).mean())
        assert np.abs(m_gpu - mom_o).max() <= (1 + step)*1e-12*scale + 2*floor
        # the drift inside the pass: bit-exact on the pass's own numbers
        assert np.array_equal(p_gpu, oracle.drift(pos_prev, m_gpu, dtm, L))
        pos_o = oracle.drift(pos_o, mom_o, dtm, L)
        d = np.abs(p_gpu - pos_o)
        d = np.minimum(d, L - d)   # a particle on the box seam may wrap on one side only
        assert d.max() <= 1e-13*L
        # from here on the oracle follows the GPU's trajectory bit for bit, so that the next
        # step's comparison measures that step alone
        pos_o, mom_o = p_gpu, m_gpu
    mesh.close()


def test_fused_pass_north_star_size_properties(torch_cuda):
    """The fused pass at the metric's own size (2^28 particles / 1024^3 mesh, thermal momenta
    as in bench.py), two consecutive steps: particle count and identities preserved, every
    particle in the region of its key, no error flags; on a strided sample of tiles the
    positions and momenta are BIT-EQUAL to cg_gather_kick_tiled + cg_drift on the same
    potential; the mesh force transfers no net momentum; the region deposit conserves mass
    (first step from the dense tile order, then from regions with gaps)."""
    _fused_pass_size_properties(torch_cuda, 1024, 2**28, 97)


def test_fused_pass_config3_size_properties(torch_cuda):
    """The same at BASELINE configs[3]'s own size on one GPU — 1024^3 = 2^30 particles on a
    2048^3 mesh (meant for 8 GPUs; ~200 GB here) — : the step bench.py times as
    configs.c3_1024c_2048."""
    torch = torch_cuda
    # (what the tests before this one left in torch's allocator and in the mesh cache goes back
    # to the device first)
    import gc
    from concept_amd import mesh as mesh_module
    gc.collect()
    mesh_module.free_meshes()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 230e9:
        pytest.skip(f'needs ~230 GB of device memory, {free/1e9:.0f} GB are free')
    _fused_pass_size_properties(torch_cuda, 2048, 2**30, 389, lean=True)


def _fused_pass_size_properties(torch, N, n, stride, lean=False):
    from concept_amd.mesh import PotentialMesh
    L = float(N)
    dt, mass = 1e-4, 1.0
    dtm, kick = dt/mass, mass*(-dt)
    mesh = PotentialMesh(N, L)
    cap = mesh.region_capacity(n)
    gen = torch.Generator(device='cuda').manual_seed(11)
    pos = torch.empty((cap, 3), dtype=torch.float64, device='cuda')
    torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen, out=pos[:n])
    pos[:n].mul_(L*(1 - 1e-13))
    mom = torch.empty((cap, 3), dtype=torch.float64, device='cuda')
    torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=gen, out=mom[:n])
    mom[:n].mul_(0.2*(L/N)*mass/dt/3**0.5)      # rms displacement 0.2 cells per step
    ids = torch.empty(cap, dtype=torch.int64, device='cuda')
    ids[:n] = torch.arange(n, device='cuda')
    pos2, mom2, ids2 = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
    table = mesh.new_tile_table()
    mesh.drift_sort(pos[:n], mom[:n], ids[:n], pos2[:n], mom2[:n], ids2[:n], dtm, table)
    pos, pos2, mom, mom2, ids, ids2 = pos2, pos, mom2, mom, ids2, ids
    start, count = table[:mesh.table_entries], None
    tabs = [mesh.new_region_table(), mesh.new_region_table()]
    C = -L**2/np.pi
    contribution = mass*float(N)**(-3)*(N/L)**3   # bench.py's (G = 1)
    for step in range(2):
        if count is None:
            mesh.deposit_tiled(pos[:n], table, contribution, accumulate=False)
        else:
            mesh.deposit_regions(pos, start, count, contribution)
        # mass conservation of the deposit from (gapped) regions
        per = mesh.layer_doubles
        rows = per//mesh.pad
        buf = torch.empty(64*per, dtype=torch.float64, device='cuda')
        tot = 0.0
        for l0 in range(0, N, 64):
            mesh.layers_read(l0, 64, buf)
            tot += float(buf.view(64, rows, mesh.pad)[:, :N, :N].sum())
        del buf
        assert abs(tot - n*contribution) <= 1e-9*n*contribution
        mesh.poisson_solve(4, C, False, 0.0)
        start_out, count_out = tabs[step]
        mesh.predict_regions(start, count, start_out)
        mesh.gather_kick_drift_scatter(pos, mom, ids, start, count, pos2, mom2, ids2, start_out,
                                       count_out, 2, kick, dtm)
        assert mesh.error_flags() == 0
        ct_out = count_out.long()
        assert int(ct_out.sum()) == n
        # --- a strided sample of input regions through the separate kernels, same potential
        st_in = start.long()
        ct_in = (st_in[1:] - st_in[:-1]) if count is None else count.long()
        nreg = ct_in.numel()
        pick = torch.arange(0, nreg, stride, device='cuda')
        lens = ct_in[pick]
        tot_s = int(lens.sum())
        offs = torch.cumsum(lens, 0) - lens
        src = torch.repeat_interleave(st_in[pick] - offs, lens) + torch.arange(tot_s,
                                                                                device='cuda')
        sp, sm, si = pos[src].contiguous(), mom[src].contiguous(), ids[src]
        mesh.gather_kick(sp, sm, 2, kick)          # direct kernel: same FD + CIC expressions
        mesh.drift(sp, sm, dtm)
        if lean:
            # (configs[3]'s size: no arrays over all slots beside the 210 GB of the step itself)
            # the sampled particles are looked for in the regions their new positions belong to
            want = _keys(torch, mesh, sp, N)
            regs = torch.unique(want)
            so, co = start_out.long()[regs], ct_out[regs]
            o2 = torch.cumsum(co, 0) - co
            slots = torch.repeat_interleave(so - o2, co) + torch.arange(int(co.sum()),
                                                                        device='cuda')
            found_ids, by = torch.sort(ids2[slots])
            at = torch.searchsorted(found_ids, si).clamp(max=found_ids.numel() - 1)
            assert torch.equal(found_ids[at], si)          # every one of them is there,
            dst = slots[by[at]]
            assert torch.equal(pos2[dst], sp) and torch.equal(mom2[dst], sm)   # bit for bit,
            # ... and everything those regions hold belongs to them
            held = torch.repeat_interleave(regs, co)
            assert torch.equal(_keys(torch, mesh, pos2[slots], N), held)
            del want, regs, slots, found_ids, by, at, dst, held, sp, sm, si, src
            pos, pos2, mom, mom2, ids, ids2 = pos2, pos, mom2, mom, ids2, ids
            start, count = start_out, count_out
            continue
        # where did those particles go?  look them up by id in the output
        live, region = _live(torch, start_out, count_out, cap)
        out_ids = ids2[live]
        assert int(live.sum()) == n
        # identities preserved: sum and sum of squares (mod 2^64) of a permutation of 0..n-1
        assert int(out_ids.sum()) == n*(n - 1)//2
        slot_of = torch.empty(n, dtype=torch.int64, device='cuda')
        slot_of[out_ids] = torch.nonzero(live).flatten()
        del out_ids
        dst = slot_of[si]
        del slot_of
        assert torch.equal(pos2[dst], sp) and torch.equal(mom2[dst], sm)
        # every sampled particle sits in the region of its key
        assert torch.equal(_keys(torch, mesh, sp, N), region[dst])
        del live, region, dst, sp, sm, si, src
        pos, pos2, mom, mom2, ids, ids2 = pos2, pos, mom2, mom, ids2, ids
        start, count = start_out, count_out
    if lean:
        mesh.close()
        return
    # a third pass from the gapped regions with the momenta at zero and no drift: what it
    # stores are the kicks themselves — the mesh force transfers no net momentum — and the
    # positions come through unchanged, in the same regions
    mesh.deposit_regions(pos, start, count, contribution)
    mesh.poisson_solve(4, C, False, 0.0)
    mom.zero_()
    start_out, count_out = mesh.new_region_table()
    mesh.predict_regions(start, count, start_out)
    mesh.gather_kick_drift_scatter(pos, mom, ids, start, count, pos2, mom2, ids2, start_out,
                                   count_out, 2, kick, 0.0)
    assert mesh.error_flags() == 0
    assert torch.equal(count_out, count)
    live, _ = _live(torch, start_out, count_out, cap)
    kicks = mom2[live]
    assert float(kicks.abs().max()) > 0
    assert float(kicks.sum(0).abs().max()) <= 1e-9*float(kicks.abs().sum(0).max())
    live_in, _ = _live(torch, start, count, cap)
    assert int(ids2[live].sum()) == n*(n - 1)//2
    assert float(pos2[live].sum()) == float(pos[live_in].sum()) or \
        abs(float(pos2[live].sum()) - float(pos[live_in].sum())) <= 1e-12*float(pos[live_in].sum())
    mesh.close()


@pytest.mark.parametrize('dist', ['uniform', 'clustered'])
def test_fused_pass_leaves_the_sum_of_mom2(torch_cuda, dist):
    """cg_set_momentum_sum: the fused pass's own sum of |mom|^2 over the momenta it leaves
    (what Timeloop's v_rms reads after every kick, analysis.py:3902-3910) against the stand-alone
    reduction cg_measure_momentum_regions over the pass's output, three steps from regions with
    gaps; the clustered box runs the heavy tiles' extra blocks (cgk_tile_order) too."""
    torch = torch_cuda
    from concept_amd.distributed import ParticleStore, RegionParticles
    from concept_amd.mesh import PotentialMesh
    N, L, n = 256, 256.0, 128**3
    mesh = PotentialMesh(N, L)
    gen = torch.Generator(device='cuda').manual_seed(11)
    pos = torch.rand((n, 3), dtype=torch.float64, device='cuda', generator=gen)*L
    if dist == 'clustered':
        blob = 128.0 + torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=gen)*4.0
        pos = torch.where(torch.rand(n, device='cuda', generator=gen)[:, None] < 0.7,
                          torch.remainder(blob, L), pos)
    pos.clamp_(0.0, L*(1 - 1e-13))
    mom = torch.randn((n, 3), dtype=torch.float64, device='cuda', generator=gen)*3.0
    store = ParticleStore(mesh, pos, mom)
    store.tile_sort()
    rp = RegionParticles(store)
    for step in range(3):
        rp.deposit(1e-3/N**3)
        mesh.poisson_solve(4, -L**2/np.pi, False, 0.0)
        rp.kick_drift_sort(2, -0.05, 0.02)
        rp.check()
        fused, nan = rp.measure_momentum(want_max=False)
        direct, biggest = rp.measure_momentum()
        assert nan != nan and biggest > 0
        assert direct > 0 and abs(fused - direct) <= 1e-13*direct, (fused, direct, step)
    assert rp.n == n
    mesh.close()
