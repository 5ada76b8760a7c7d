"""concept_amd.distributed — the PM path sharded over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).
Decomposition: x-SLAB DOMAINS for everything.  Rank r owns the
mesh layers x in [r*N/P, (r+1)*N/P) and the particles whose lower CIC cell
lies there.  This replaces the reference's pair of decompositions — 3-D
domains for particles/real-space grids, x-slabs for the FFT
(communication.py:692-741, mesh.py:1935-1942) — so its slab<->domain remaps
(mesh.py:2138-2411) vanish, every halo is a contiguous block of mesh layers
exchanged with the two ring neighbours, and the only all-to-all is the FFT
transpose.  Results do not depend on the decomposition beyond summation order
(the reference's own bar: test/nprocs_pm/analyze.py:121, 1e-9).

Exchange steps per PM kick (what replaces which MPI call site):
  deposit ghost fold   communicate_ghosts(grid,'+=')  1 layer  -> next rank, added
  FFT transpose        FFTW-MPI alltoall (fft.c:240)  all_to_all_single, twice
  potential ghosts     communicate_ghosts(grid,'=')   G layers <-> both neighbours
  particle exchange    exchange() (communication.py:135-517) after each drift
The pack/unpack of the transpose is fused into the y pass of the FFT
(cg_fft.hip), the halos need no packing at all.

The same code runs under the "gloo" backend (tests: two ranks sharing one GPU,
or CPU-only checks of the exchange logic) by staging messages through host
memory; that path is for tests only.
"""
import os

import torch
import torch.distributed as dist

from . import lib
from .mesh import PotentialMesh

DEAD_X_FACTOR = -4.0  # pos.x = DEAD_X_FACTOR*boxsize marks a vacated particle slot


class Comm:
    """Thin wrapper over torch.distributed that also works on gloo."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.stage = self.backend != 'nccl'  # gloo: stage device tensors through the host

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if not self.stage:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
            return
        # gloo: pairwise exchange of the blocks through host memory
        P = self.world
        src = inp.cpu()
        if in_splits is None:
            in_splits = [src.shape[0]//P]*P
            out_splits = [out.shape[0]//P]*P
        ichunks = list(torch.split(src, in_splits))
        res = torch.empty(out.shape, dtype=out.dtype)
        ochunks = list(torch.split(res, out_splits))
        ops = []
        for q in range(P):
            if q == self.rank:
                ochunks[q].copy_(ichunks[q])
            else:
                if ichunks[q].numel():
                    ops.append(dist.P2POp(dist.isend, ichunks[q].contiguous(), q,
                                          group=self.group))
                if ochunks[q].numel():
                    ops.append(dist.P2POp(dist.irecv, ochunks[q], q, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        out.copy_(res)

    def all_to_all_layers(self, out, inp, nlayers_total, layer0, nlayers, async_op=False):
        """all_to_all of the layers [layer0, layer0 + nlayers) of every peer block of two
        transpose buffers (P blocks of `nlayers_total` layers each: the range is one contiguous
        piece per peer, exchanged in place).  With async_op the RCCL work handle is returned:
        the exchange runs on RCCL's stream behind what the current stream has queued so far,
        and `wait()` makes the current stream wait for it."""
        P = self.world
        o = out.view(P, nlayers_total, -1)[:, layer0:layer0 + nlayers]
        i = inp.view(P, nlayers_total, -1)[:, layer0:layer0 + nlayers]
        if not self.stage:
            return dist.all_to_all([o[q] for q in range(P)], [i[q] for q in range(P)],
                                   group=self.group, async_op=async_op)
        # gloo (tests): pairwise through host memory, synchronously
        ops, recvs = [], {}
        for q in range(P):
            if q == self.rank:
                o[q].copy_(i[q])
                continue
            ops.append(dist.P2POp(dist.isend, i[q].cpu().contiguous(), q, group=self.group))
            recvs[q] = torch.empty(o[q].shape, dtype=o.dtype)
            ops.append(dist.P2POp(dist.irecv, recvs[q], q, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for q, r in recvs.items():
            o[q].copy_(r)
        return None

    def sendrecv(self, send, dest, recv, source):
        """send -> dest while receiving <- source (a ring shift)."""
        if dest == self.rank and source == self.rank:
            recv.copy_(send)
            return
        if self.stage:
            s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        else:
            s, r = send, recv
        ops = []  # empty messages are skipped on both ends (sizes are known to both)
        if s.numel():
            ops.append(dist.P2POp(dist.isend, s, dest, group=self.group))
        if r.numel():
            ops.append(dist.P2POp(dist.irecv, r, source, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if self.stage and r.numel():
            recv.copy_(r)

    def all_gather_ints(self, values):
        t = torch.tensor(values, dtype=torch.int64)
        if not self.stage:
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.stack(out).cpu()


class SlabDomain:
    """The mesh side of one rank: local mesh slab + exchange buffers."""

    def __init__(self, gridsize, boxsize, nghosts=2, device=None, group=None):
        self.comm = Comm(group)
        self.rank, self.world = self.comm.rank, self.comm.world
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.mesh = PotentialMesh(gridsize, boxsize, nghosts=nghosts, device=self.device,
                                  nprocs=self.world, rank=self.rank)
        m = self.mesh
        self.N, self.nxl, self.G = m.gridsize, m.nxl, m.ghost_layers
        self.boxsize = float(boxsize)
        per = m.layer_doubles
        self._layer = per
        self.tbuf_a = torch.empty(m.transpose_doubles, dtype=torch.float64, device=self.device)
        self.tbuf_b = torch.empty(m.transpose_doubles, dtype=torch.float64, device=self.device)
        self.halo_s = torch.empty(self.G*per, dtype=torch.float64, device=self.device)
        self.halo_r = torch.empty(self.G*per, dtype=torch.float64, device=self.device)
        self.next = (self.rank + 1) % self.world
        self.prev = (self.rank - 1) % self.world
        # The FFT transposes are exchanged in `pieces` layer ranges so that the transform of one
        # range overlaps the exchange of the previous one (CONCEPT_GPU_DIST_PIECES, 1 = one
        # all_to_all_single per transpose).  A piece should stay a large message: >= 8 layers.
        # CONCEPT_GPU_DIST_FORCE=1: a single rank also takes the transposing solve (tests)
        self.force_dist = os.environ.get('CONCEPT_GPU_DIST_FORCE') == '1'
        want = int(os.environ.get('CONCEPT_GPU_DIST_PIECES', '4'))
        npieces = max(1, min(want, self.nxl//8))
        if npieces > 1:
            # a piece should also fit the 256 MB infinity cache, like the chunks of the
            # single-GPU schedule (cg_fft.hip zy_chunk_layers): its y pass then reads what its
            # z pass wrote from the cache
            cache_layers = max(1, int(266e6//(per*8)))
            npieces = min(max(npieces, -(-self.nxl//cache_layers)), max(1, self.nxl//8))
        if npieces > 1 and not self.comm.stage:
            # every rank probes the asynchronous list form of all_to_all once, on a few bytes;
            # a transport that rejects it falls back to one all_to_all_single per transpose
            try:
                a = torch.zeros(2*self.world, dtype=torch.float64, device=self.device)
                b = torch.empty_like(a)
                w = self.comm.all_to_all_layers(b, a, 2, 0, 1, async_op=True)
                if w is not None:
                    w.wait()
                torch.cuda.synchronize(self.device)
            except Exception as e:  # noqa: BLE001 (any backend error means: do not pipeline)
                print(f'[concept_amd] pipelined transposes disabled: {e}', flush=True)
                npieces = 1
        edges = [self.nxl*k//npieces for k in range(npieces + 1)]
        self.pieces = [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]

    # communicate_ghosts(grid, '+=') after the deposit (mesh.py:609)
    def fold_deposit_ghost(self):
        if self.world == 1:
            return  # one periodic domain: the deposit wraps by itself, no ghost layers
        per = self._layer
        s, r = self.halo_s[:per], self.halo_r[:per]
        self.mesh.layers_read(self.nxl, 1, s)
        self.comm.sendrecv(s, self.next, r, self.prev)
        self.mesh.layers_write(0, 1, r, add=True)

    # communicate_ghosts(grid, '=') of the potential (interactions.py:2303-2307)
    def fill_potential_ghosts(self):
        if self.world == 1:
            return
        G = self.G
        # my first G layers -> previous rank's upper ghosts [nxl, nxl+G)
        self.mesh.layers_read(0, G, self.halo_s)
        self.comm.sendrecv(self.halo_s, self.prev, self.halo_r, self.next)
        self.mesh.layers_write(self.nxl, G, self.halo_r, add=False)
        # my last G layers -> next rank's lower ghosts [-G, 0)
        self.mesh.layers_read(self.nxl - G, G, self.halo_s)
        self.comm.sendrecv(self.halo_s, self.next, self.halo_r, self.prev)
        self.mesh.layers_write(-G, G, self.halo_r, add=False)

    # A3..A8 with the transpose (fft.c:240-257) as all_to_all_single
    def poisson_solve(self, deconv_order, C, long_range=False, E=0.0):
        m = self.mesh
        if self.world == 1 and not self.force_dist:
            m.poisson_solve(deconv_order, C, long_range, E)
            return
        if len(self.pieces) == 1:
            m.dist_fft_forward(self.tbuf_a)
            self.comm.all_to_all(self.tbuf_b, self.tbuf_a)
            m.dist_fft_xsolve(self.tbuf_b, deconv_order, C, long_range, E)
            self.comm.all_to_all(self.tbuf_a, self.tbuf_b)
            m.dist_fft_backward(self.tbuf_a)
            return
        # pipelined: z + y transform of layer range k+1 while range k is on the links ...
        works = []
        for l0, nl in self.pieces:
            m.dist_fft_forward(self.tbuf_a, l0, nl)
            works.append(self.comm.all_to_all_layers(self.tbuf_b, self.tbuf_a, self.nxl, l0, nl,
                                                     async_op=True))
        for w in works:
            if w is not None:
                w.wait()
        m.dist_fft_xsolve(self.tbuf_b, deconv_order, C, long_range, E)
        # ... and on the way back the inverse y + z of range k while range k+1 arrives
        works = [self.comm.all_to_all_layers(self.tbuf_a, self.tbuf_b, self.nxl, l0, nl,
                                             async_op=True) for l0, nl in self.pieces]
        for (l0, nl), w in zip(self.pieces, works):
            if w is not None:
                w.wait()
            m.dist_fft_backward(self.tbuf_a, l0, nl)


class DistributedParticles:
    """Particles of one rank: arrays with spare capacity, n valid entries."""

    def __init__(self, domain, pos, mom, ids=None, slack=1.3):
        self.domain = domain
        n = pos.shape[0]
        cap = int(n*slack) + 1024
        dev = domain.device
        self.cap = cap
        self.n = n
        self.pos = torch.zeros((cap, 3), dtype=torch.float64, device=dev)
        self.mom = torch.zeros((cap, 3), dtype=torch.float64, device=dev)
        self.ids = torch.zeros(cap, dtype=torch.int64, device=dev)
        self.pos[:n] = pos
        self.mom[:n] = mom
        # without ids the sort moves pos and mom only (the id column of the exchanged rows is
        # carried but meaningless)
        self.has_ids = ids is not None
        if ids is not None:
            self.ids[:n] = ids
        self.pos2 = torch.empty_like(self.pos)
        self.mom2 = torch.empty_like(self.mom)
        self.ids2 = torch.empty_like(self.ids)
        self.table = domain.mesh.new_tile_table()
        self.sorted = False
        # rows that the drift prepared by the last pm_kick(next_dt_over_mass=...) takes out of the
        # slab, listed by the gather-kick itself (cg_set_emigrant_list)
        self.emig_idx = torch.empty(max(4096, cap//16), dtype=torch.int64, device=dev)
        self.emig_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self._emig_for = None  # (pos pointer, n, dt_over_mass) the list was made for
        self.emigrants_total = 0  # rows this rank has shipped so far (all exchanges)

    def check(self):
        """Raise if a kernel recorded an inconsistency since the last call (synchronises)."""
        self.domain.mesh.check_errors()

    def touch_mom(self):
        """Call after changing `mom` by anything but pm_kick: the drift histogram and the
        emigrant list the last kick prepared no longer describe the particles."""
        self._emig_for = None
        self.domain.mesh.prepare_invalidate()

    def view(self, name):
        return getattr(self, name)[:self.n]

    def drift(self, dt_over_mass):
        self.domain.mesh.drift(self.view('pos'), self.view('mom'), dt_over_mass)
        self.sorted = False

    def exchange(self):
        """exchange() (communication.py:135-517): re-home particles whose lower CIC
        cell left this rank's slab."""
        d = self.domain
        owner = d.mesh.owner_rank(self.view('pos'))
        self.n, self._expected = exchange_rows(
            d.comm, owner, self.pos, self.mom, self.ids, self.n, self.cap,
            DEAD_X_FACTOR*d.boxsize)
        self.sorted = False

    def drift_exchange_sort(self, dt_over_mass):
        """drift + exchange + tile sort in fused form: the particles whose DRIFTED position
        leaves this rank's slab are shipped first (undrifted rows; the receiver drifts them
        with the same arithmetic), then one fused drift + sort pass pair runs over the
        compacted local set.  When the previous pm_kick prepared the tile histogram of this
        drift (next_dt_over_mass), the immigrants' keys are added to it and the sort needs
        no histogram pass.  Same result as drift(); exchange(); tile_sort()."""
        d = self.domain
        m = d.mesh
        pos, mom = self.view('pos'), self.view('mom')
        move = None
        if self._emig_for == (self.pos.data_ptr(), self.n, float(dt_over_mass)):
            cnt = int(self.emig_count.item())
            if cnt <= self.emig_idx.numel():  # else the list overflowed: find them the long way
                move_idx = self.emig_idx[:cnt]
                move = (move_idx, m.owner_rank_drifted(pos[move_idx], mom[move_idx],
                                                       dt_over_mass))
        self._emig_for = None
        owner = m.owner_rank_drifted(pos, mom, dt_over_mass) if move is None else None
        n_new, inc = exchange_rows_compact(d.comm, owner, self.pos, self.mom, self.ids, self.n,
                                           self.cap, move=move)
        self.emigrants_total += d.comm.last_sent
        self.n = n_new
        pos, mom = self.view('pos'), self.view('mom')
        if inc is not None and inc.shape[0]:
            m.prepare_rebind(pos, mom, inc[:, 0:3].contiguous(), inc[:, 3:6].contiguous())
        else:
            m.prepare_rebind(pos, mom)
        m.drift_sort(pos, mom, self.view('ids') if self.has_ids else None, self.pos2[:n_new],
                     self.mom2[:n_new], self.ids2[:n_new] if self.has_ids else None, dt_over_mass,
                     self.table)
        self.pos, self.pos2 = self.pos2, self.pos
        self.mom, self.mom2 = self.mom2, self.mom
        self.ids, self.ids2 = self.ids2, self.ids
        kept = int(self.table[-1].item()) & 0xffffffff
        if kept != n_new:
            raise lib.ConceptGPUError(
                f'rank {d.rank}: fused drift + sort kept {kept} of {n_new} particles')
        self._expected = kept
        self.sorted = True

    def tile_sort(self):
        d = self.domain
        d.mesh.sort_particles(self.view('pos'), self.view('mom'),
                              self.view('ids') if self.has_ids else None,
                              self.pos2[:self.n], self.mom2[:self.n],
                              self.ids2[:self.n] if self.has_ids else None, self.table)
        self.pos, self.pos2 = self.pos2, self.pos
        self.mom, self.mom2 = self.mom2, self.mom
        self.ids, self.ids2 = self.ids2, self.ids
        kept = int(self.table[-1].item()) & 0xffffffff
        expected = getattr(self, '_expected', self.n)
        if kept != expected:
            raise lib.ConceptGPUError(
                f'rank {d.rank}: tile sort kept {kept} of {expected} particles — particles '
                'outside this rank\'s slab (exchange() must run after every drift)')
        self.n = kept
        self._expected = kept
        self.sorted = True


def exchange_rows(comm, owner, pos, mom, ids, n, cap, dead_x):
    """Move the particles with owner != comm.rank to their owners (all-to-all-v of
    rows pos(3) mom(3) id(1)).  Vacated slots are refilled with immigrants, surplus
    immigrants are appended after slot n, leftover holes get pos.x = dead_x (the tile
    sort drops them).  Works on any device.  Returns (n_slots, n_alive)."""
    P, rank = comm.world, comm.rank
    dev = pos.device
    move_idx = torch.nonzero(owner[:n] != rank).flatten()
    dest = owner[move_idx].long()
    order = torch.argsort(dest, stable=True)
    move_idx, dest = move_idx[order], dest[order]
    send_counts = torch.bincount(dest, minlength=P).cpu().tolist()
    rows = torch.empty((move_idx.numel(), 7), dtype=torch.float64, device=dev)
    rows[:, 0:3] = pos[move_idx]
    rows[:, 3:6] = mom[move_idx]
    rows[:, 6] = ids[move_idx].view(torch.float64)
    counts = comm.all_gather_ints(send_counts)  # counts[src][dst]
    recv_counts = counts[:, rank].tolist()
    m_in, m_out = int(sum(recv_counts)), int(move_idx.numel())
    inc = torch.empty((m_in, 7), dtype=torch.float64, device=dev)
    comm.all_to_all(inc, rows, recv_counts, send_counts)
    k = min(m_in, m_out)
    if k:
        h = move_idx[:k]
        pos[h] = inc[:k, 0:3]
        mom[h] = inc[:k, 3:6]
        ids[h] = inc[:k, 6].contiguous().view(torch.int64)
    n_slots = n
    if m_in > k:
        extra = m_in - k
        if n + extra > cap:
            raise lib.ConceptGPUError(
                f'rank {rank}: particle capacity {cap} exceeded ({n + extra})')
        pos[n:n + extra] = inc[k:, 0:3]
        mom[n:n + extra] = inc[k:, 3:6]
        ids[n:n + extra] = inc[k:, 6].contiguous().view(torch.int64)
        n_slots = n + extra
    elif m_out > k:
        pos[move_idx[k:], 0] = dead_x
    return n_slots, n - m_out + m_in


def exchange_rows_compact(comm, owner, pos, mom, ids, n, cap, move=None):
    """exchange_rows that leaves no dead rows: vacated slots are refilled with immigrants,
    surplus immigrants are appended, leftover holes are closed with live rows from the tail.
    `move` = (row numbers, their owners) when the leaving rows are already known (then `owner`
    is not needed).  Returns (n_new, immigrant rows (m_in, 7) or None); the live rows are
    [0, n_new)."""
    P, rank = comm.world, comm.rank
    dev = pos.device
    if move is None:
        move_idx = torch.nonzero(owner[:n] != rank).flatten()
        dest = owner[move_idx].long()
    else:
        move_idx, dest = move[0], move[1].long()
    order = torch.argsort(dest, stable=True)
    move_idx, dest = move_idx[order], dest[order]
    send_counts = torch.bincount(dest, minlength=P).cpu().tolist()
    rows = torch.empty((move_idx.numel(), 7), dtype=torch.float64, device=dev)
    rows[:, 0:3] = pos[move_idx]
    rows[:, 3:6] = mom[move_idx]
    rows[:, 6] = ids[move_idx].view(torch.float64)
    counts = comm.all_gather_ints(send_counts)  # counts[src][dst]
    recv_counts = counts[:, rank].tolist()
    m_in, m_out = int(sum(recv_counts)), int(move_idx.numel())
    comm.last_sent = m_out
    inc = torch.empty((m_in, 7), dtype=torch.float64, device=dev)
    comm.all_to_all(inc, rows, recv_counts, send_counts)
    k = min(m_in, m_out)
    if k:
        h = move_idx[:k]
        pos[h] = inc[:k, 0:3]
        mom[h] = inc[:k, 3:6]
        ids[h] = inc[:k, 6].contiguous().view(torch.int64)
    n_new = n - m_out + m_in
    if m_in > k:
        if n_new > cap:
            raise lib.ConceptGPUError(f'rank {rank}: particle capacity {cap} exceeded ({n_new})')
        pos[n:n_new] = inc[k:, 0:3]
        mom[n:n_new] = inc[k:, 3:6]
        ids[n:n_new] = inc[k:, 6].contiguous().view(torch.int64)
    elif m_out > k:
        holes = move_idx[k:]
        low = holes[holes < n_new]                       # holes to fill
        tail = torch.ones(n - n_new, dtype=torch.bool, device=dev)
        tail[holes[holes >= n_new] - n_new] = False      # tail rows that are holes themselves
        src = torch.nonzero(tail).flatten() + n_new      # live rows of the tail
        pos[low] = pos[src]
        mom[low] = mom[src]
        ids[low] = ids[src]
    return n_new, (inc if m_in else None)


def shortrange_kick(domain, particles, *, scale, range_, tilesize, tablesize, softening,
                    factor, kernel='spline'):
    """P3M short-range kick of a particle set onto itself over x-slab domains
    (component_component + sendrecv_component, interactions.py:122-329,
    communication.py:847-1130).  Each rank receives the positions of the neighbour
    ranks' particles that lie within the force range of its slab ("supplier
    particles in the boundary tiles") and runs the one-sided tile sweep for its own
    particles; because the sweep is one-sided no Δmom travels back (the reference
    returns it because its pair update is symmetric).  Returns Δmom (n, 3)."""
    from . import commons, shortrange
    d = domain
    comm, P, rank = d.comm, d.world, d.rank
    L, N = d.boxsize, d.N
    nt = int((L/1)/tilesize*(1 + commons.machine_ϵ))  # global tiling, species.py:3943-3950
    if nt < 4:
        raise lib.ConceptGPUError(
            'The global gravity tiling needs to have at least 4 tiles across the box in every '
            'direction (species.py:3971)')
    cell = L/N
    slab_w = d.nxl*cell
    if range_*1.001 >= slab_w or (P == 2 and 2.002*range_ >= slab_w):
        raise lib.ConceptGPUError('short-range force range too large for the slab width')
    pos = particles.view('pos')
    n = pos.shape[0]
    # slab in position space: lower CIC cell in [x0, x0 + nxl)  <=>  x in [xlo, xhi) (wrapped)
    xlo = (d.mesh.x0 + 0.5)*cell
    x = pos[:, 0]
    rel = torch.remainder(x - xlo, L)           # 0 .. slab_w for owned particles
    margin = range_*(1 + 1e-9) + 1e-9*L
    to_prev = pos[rel < margin]                  # near my lower face -> previous rank
    to_next = pos[rel >= slab_w - margin]        # near my upper face -> next rank

    def ship(send, dest, source):
        cnt = torch.tensor([send.shape[0]], dtype=torch.int64, device=send.device)
        got = torch.empty_like(cnt)
        comm.sendrecv(cnt, dest, got, source)
        recv = torch.empty((int(got.item()), 3), dtype=torch.float64, device=send.device)
        comm.sendrecv(send.contiguous(), dest, recv, source)
        return recv
    if P == 1:
        ghosts = [pos.new_zeros((0, 3))]
    else:
        from_next = ship(to_prev, d.prev, d.next)   # what my next rank has near ITS lower face
        from_prev = ship(to_next, d.next, d.prev)
        ghosts = [from_prev, from_next]
    supp = torch.cat([pos] + ghosts).contiguous()
    m = d.mesh
    ext = L/nt
    cells_r = m.shortrange_build(pos.contiguous(), nt, ext)
    cells_s = m.shortrange_build(supp, nt, ext)
    table, maxr2 = shortrange.get_shortrange_table(softening, scale, range_, tablesize, kernel,
                                                   pos.device)
    dmom = torch.zeros((n, 3), dtype=torch.float64, device=pos.device)
    # `same`: supplier rows 0..n-1 ARE the receivers (same order), ghosts follow
    m.shortrange_sweep(pos.contiguous(), cells_r, dmom, supp, cells_s, nt, True, table,
                       (tablesize - 1)/maxr2, range_**2, factor)
    return dmom


def pm_kick(domain, particles, contribution, deconv_order, C, kick_factor, diff_order=2,
            long_range=False, E=0.0, next_dt_over_mass=None, mark=None):
    """One long-range PM kick of a tile-sorted particle set onto itself, sharded
    (particle_mesh(), interactions.py:1985-2335).  `mark(name)` is called after each stage
    (bench.py records an event there)."""
    if not particles.sorted:
        raise lib.ConceptGPUError('pm_kick: particles must be exchanged and tile-sorted')
    mark = mark or (lambda name: None)
    m = domain.mesh
    m.deposit_tiled(particles.view('pos'), particles.table, contribution, accumulate=False)
    domain.fold_deposit_ghost()
    mark('deposit+ghost_fold')
    domain.poisson_solve(deconv_order, C, long_range, E)
    mark('poisson+transposes')
    domain.fill_potential_ghosts()
    if next_dt_over_mass is None:
        m.gather_kick_tiled(particles.view('pos'), particles.view('mom'), particles.table,
                            diff_order, kick_factor)
    else:
        # also histogram the tile keys after the NEXT drift and list the rows it takes out of the
        # slab (for drift_exchange_sort)
        m.set_emigrant_list(particles.emig_idx, particles.emig_count)
        particles._emig_for = (particles.pos.data_ptr(), particles.n, float(next_dt_over_mass))
        try:
            m.gather_kick_tiled_prepare(particles.view('pos'), particles.view('mom'),
                                        particles.table, diff_order, kick_factor,
                                        next_dt_over_mass)
        finally:
            m.set_emigrant_list(None, None)  # the launch holds the pointers; the context must not
    mark('ghost_fill+gather_kick')
