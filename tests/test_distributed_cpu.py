"""world_size-2 (and 4) gloo tests of the multi-GPU exchange logic on CPU:
the Comm wrapper, the particle exchange bookkeeping and the layout algebra of
the FFT transpose (what the fused pack in cg_fft.hip writes / reads)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(fn, world, *args):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn, args), nprocs=world, join=True)


def _entry(rank, world, port, fn, args):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fn(rank, world, *args)
    finally:
        dist.destroy_process_group()


def _w_transpose(rank, world, N):
    """send[q][i_local][j_in_block][kk] -> all-to-all -> [i_global][j_local][kk]."""
    from concept_amd.distributed import Comm
    comm = Comm()
    cp = N//2 + 1
    nxl = JB = N//world
    rng = np.random.default_rng(0)
    full = rng.normal(size=(N, N, cp))  # the global (i, j, kk) array, same on every rank
    local = full[rank*nxl:(rank + 1)*nxl]  # my slab [i_local][j][kk]
    send = np.empty((world, nxl, JB, cp))
    for q in range(world):
        send[q] = local[:, q*JB:(q + 1)*JB, :]
    recv = torch.empty(world*nxl*JB*cp, dtype=torch.float64)
    comm.all_to_all(recv, torch.from_numpy(send.reshape(-1)))
    got = recv.numpy().reshape(N, JB, cp)  # [q*nxl + i_local][j_local][kk]
    assert np.array_equal(got, full[:, rank*JB:(rank + 1)*JB, :])
    # and back
    back = torch.empty_like(recv)
    comm.all_to_all(back, recv)
    assert np.array_equal(back.numpy().reshape(world, nxl, JB, cp), send)


def _w_transpose_pieces(rank, world, N, npieces):
    """The same transpose exchanged in layer ranges (Comm.all_to_all_layers): the pieces
    together are the one all-to-all."""
    from concept_amd.distributed import Comm
    comm = Comm()
    cp = N//2 + 1
    nxl = JB = N//world
    rng = np.random.default_rng(0)
    full = rng.normal(size=(N, N, cp))
    local = full[rank*nxl:(rank + 1)*nxl]
    send = np.empty((world, nxl, JB, cp))
    for q in range(world):
        send[q] = local[:, q*JB:(q + 1)*JB, :]
    send_t = torch.from_numpy(send.reshape(-1).copy())
    recv = torch.full((world*nxl*JB*cp,), np.nan, dtype=torch.float64)
    edges = [nxl*k//npieces for k in range(npieces + 1)]
    for a, b in zip(edges[:-1], edges[1:]):
        w = comm.all_to_all_layers(recv, send_t, nxl, a, b - a, async_op=True)
        if w is not None:
            w.wait()
    assert np.array_equal(recv.numpy().reshape(N, JB, cp), full[:, rank*JB:(rank + 1)*JB, :])
    back = torch.full_like(recv, np.nan)
    for a, b in reversed(list(zip(edges[:-1], edges[1:]))):  # any order
        comm.all_to_all_layers(back, recv, nxl, a, b - a)
    assert np.array_equal(back.numpy().reshape(world, nxl, JB, cp), send)


def _w_ring(rank, world):
    from concept_amd.distributed import Comm
    comm = Comm()
    s = torch.full((5,), float(rank), dtype=torch.float64)
    r = torch.empty(5, dtype=torch.float64)
    comm.sendrecv(s, (rank + 1) % world, r, (rank - 1) % world)
    assert (r == float((rank - 1) % world)).all()
    comm.sendrecv(s, (rank - 1) % world, r, (rank + 1) % world)
    assert (r == float((rank + 1) % world)).all()


@pytest.mark.parametrize('world', [2, 4])
def test_transpose_layout(world):
    _run(_w_transpose, world, 16)


def test_ring_sendrecv():
    _run(_w_ring, 2)
    _run(_w_ring, 3)



def _w_exchange_compact(rank, world, n_per, skew, listed=False):
    """exchange_rows_compact: afterwards the live rows are exactly [0, n_new), every particle
    is at home, rows stayed together, nothing lost or duplicated; `skew` makes one rank
    lose far more than it gains (holes closed from the tail) and another gain more."""
    from concept_amd.distributed import Comm, exchange_rows_compact
    comm = Comm()
    gen = torch.Generator().manual_seed(200 + rank)
    cap = 4*n_per
    pos = torch.zeros((cap, 3), dtype=torch.float64)
    mom = torch.zeros((cap, 3), dtype=torch.float64)
    ids = torch.zeros(cap, dtype=torch.int64)
    n = n_per + 11*rank
    x = torch.rand((n, 3), dtype=torch.float64, generator=gen)
    if skew:  # almost everything of rank 0 belongs to the last rank
        x[:, 0] = x[:, 0]**(0.15 if rank == 0 else 1.0)
    pos[:n] = x
    mom[:n] = pos[:n]*3 + 1
    ids[:n] = torch.arange(n) + 10**6*rank
    pos[n:] = float('nan')  # anything beyond n must never be picked up
    owner = torch.clamp((pos[:n, 0]*world).long(), max=world - 1).int()
    before = comm.all_gather_ints([n])[:, 0].sum().item()
    if listed:  # the leaving rows come as a list in arbitrary order (the gather-kick's)
        idx = torch.nonzero(owner != rank).flatten()
        idx = idx[torch.randperm(idx.numel(), generator=gen)]
        n_new, inc = exchange_rows_compact(comm, None, pos, mom, ids, n, cap,
                                           move=(idx, owner[idx]))
    else:
        n_new, inc = exchange_rows_compact(comm, owner, pos, mom, ids, n, cap)
    p, m, i = pos[:n_new], mom[:n_new], ids[:n_new]
    assert not torch.isnan(p).any()
    assert (torch.clamp((p[:, 0]*world).long(), max=world - 1) == rank).all()
    assert torch.equal(m, p*3 + 1)
    m_in = 0 if inc is None else inc.shape[0]
    if m_in:  # the immigrant rows are among the live rows
        live = {tuple(r) for r in p.tolist()}
        assert all(tuple(r) in live for r in inc[:, 0:3].tolist())
    after = comm.all_gather_ints([n_new])[:, 0].sum().item()
    assert after == before
    all_ids = [None]*world
    dist.all_gather_object(all_ids, i.tolist())
    flat = sorted(v for l in all_ids for v in l)
    assert len(flat) == len(set(flat)) == before


@pytest.mark.parametrize('world,skew,listed', [(2, False, False), (4, False, False),
                                               (3, True, False), (3, True, True),
                                               (4, False, True)])
def test_particle_exchange_compact(world, skew, listed):
    _run(_w_exchange_compact, world, 400, skew, listed)


@pytest.mark.parametrize('world,N,npieces', [(2, 16, 2), (4, 32, 4), (2, 16, 3)])
def test_transpose_in_layer_pieces(world, N, npieces):
    _run(_w_transpose_pieces, world, N, npieces)


def _w_floats_and_snapshot_shares(rank, world):
    """Comm.all_gather_floats: the same array on every rank, in rank order (what the v_rms of
    the time loop sums over); and the rank-wise snapshot reader: every rank its own rows of a
    reference-written GADGET file, together the whole file (communication.partition)."""
    from concept_amd import commons, snapshot
    from concept_amd.distributed import Comm
    comm = Comm()
    got = comm.all_gather_floats([rank + 0.25, 10.0*rank])
    want = np.array([[r + 0.25, 10.0*r] for r in range(world)])
    assert np.array_equal(got.numpy(), want)
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, 'golden', 'gadget_sf2_32.npz'))
    commons.load_params({'boxsize': float(g['boxsize'])})
    path = os.path.join(here, 'golden', 'gadget_sf2_32.gadget')
    mine = snapshot.load(path, rank=rank, nprocs=world)
    whole = snapshot.load(path, rank=0, nprocs=1)
    for c, w in zip(mine.components, whole.components):
        counts = comm.all_gather_ints([c['N_local'], c['start_local']])
        assert int(counts[:, 0].sum()) == w['N']
        assert np.array_equal(c['pos'], w['pos'][c['start_local']:c['start_local'] + c['N_local']])
        assert np.array_equal(c['mom'], w['mom'][c['start_local']:c['start_local'] + c['N_local']])


@pytest.mark.parametrize('world', [2, 3])
def test_gather_floats_and_rank_wise_snapshot(world):
    _run(_w_floats_and_snapshot_shares, world)
