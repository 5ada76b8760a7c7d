"""concept_amd.distributed — particles and the PM kick on x-slab domains.

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
The mesh side lives in concept_amd.mesh.PotentialMesh (comm=...); this module
holds the particle side: ParticleStore (capacity arrays of any set of
per-particle columns, tile sort, exchange()) and the bench's raw PM kick.

The host waits for the GPU ONCE per step: the gather-kick lists the rows its
prepared drift takes out of the slab, a small kernel counts them per
destination, the counts are exchanged on the device and copied to pinned
memory; the next step starts by waiting for that copy, after which every
message size is known and the whole step is enqueued without another round
trip.
"""
import ctypes
import os

import torch

from . import lib
from .comm import Comm
from .mesh import PotentialMesh



def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


class SlabDomain:
    """One rank's mesh slab with its exchange buffers: a PotentialMesh bound to a process
    group (kept as the handle bench.py and the tests use)."""

    def __init__(self, gridsize, boxsize, nghosts=2, device=None, group=None):
        self.comm = Comm(group)
        self.comm.force = os.environ.get('CONCEPT_GPU_DIST_FORCE') == '1'
        self.rank, self.world = self.comm.rank, self.comm.world
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.mesh = PotentialMesh(gridsize, boxsize, nghosts=nghosts, device=self.device,
                                  comm=self.comm)
        m = self.mesh
        self.N, self.nxl, self.G = m.gridsize, m.nxl, m.ghost_layers
        self.boxsize = float(boxsize)
        self.force_dist = self.comm.force
        self.next, self.prev = self.comm.next, self.comm.prev

    tbuf_a = property(lambda self: self.mesh.stage)
    tbuf_b = property(lambda self: self.mesh.four)
    pieces = property(lambda self: self.mesh.pieces)

    def fold_deposit_ghost(self):
        self.mesh.fold_ghosts()

    def fill_potential_ghosts(self):
        self.mesh.fill_ghosts()

    def poisson_solve(self, deconv_order, C, long_range=False, E=0.0):
        self.mesh.poisson_solve(deconv_order, C, long_range, E)


# ---------------------------------------------------------------------------
# rows: the per-particle columns of a store packed into float64 rows for the wire
# ---------------------------------------------------------------------------
def _row_width(cols):
    return sum((c.shape[1] if c.dim() == 2 else 1) for c in cols)


def _pack_rows(cols, idx):
    """(len(idx), W) float64: the rows `idx` of every column side by side (int64 travels as
    its bit pattern, the int8 rung indices as exact small doubles)."""
    rows = torch.empty((idx.numel(), _row_width(cols)), dtype=torch.float64, device=idx.device)
    o = 0
    for c in cols:
        v = c[idx]
        if c.dim() == 2:
            rows[:, o:o + c.shape[1]] = v
            o += c.shape[1]
            continue
        if c.dtype == torch.int64:
            rows[:, o] = v.view(torch.float64)
        else:
            rows[:, o] = v.to(torch.float64)
        o += 1
    return rows


def _unpack_rows(cols, rows, where):
    """cols[where] = rows (where: index tensor or slice)"""
    o = 0
    for c in cols:
        if c.dim() == 2:
            c[where] = rows[:, o:o + c.shape[1]]
            o += c.shape[1]
            continue
        v = rows[:, o].contiguous()
        c[where] = v.view(torch.int64) if c.dtype == torch.int64 else v.to(c.dtype)
        o += 1


def exchange_columns(comm, cols, n, cap, move_idx, dest, send_counts, recv_counts):
    """The data movement of exchange() (communication.py:135-517) once every size is known on
    the host: the rows `move_idx` (device int64, any order) go to the ranks `dest` (device,
    same length); send_counts / recv_counts are host lists.  Vacated slots are refilled with
    immigrants, surplus immigrants are appended, leftover holes are closed with live rows from
    the tail: afterwards the live rows are exactly [0, n_new).  No host round trip.
    Returns (n_new, immigrant rows (m_in, W) float64 or None)."""
    rank = comm.rank
    m_out, m_in = int(sum(send_counts)), int(sum(recv_counts))
    if m_out != move_idx.numel() or send_counts[rank] or recv_counts[rank]:
        raise lib.ConceptGPUError('exchange_columns: inconsistent message sizes')
    dev = cols[0].device
    order = torch.argsort(dest.long(), stable=True)
    move_idx = move_idx[order]
    rows = _pack_rows(cols, move_idx)
    inc = torch.empty((m_in, rows.shape[1]), dtype=torch.float64, device=dev)
    comm.all_to_all(inc, rows, list(recv_counts), list(send_counts))
    comm.last_sent = m_out
    n_new = n - m_out + m_in
    if n_new > cap:
        raise lib.ConceptGPUError(f'rank {rank}: particle capacity {cap} exceeded ({n_new})')
    k = min(m_in, m_out)
    if k:
        _unpack_rows(cols, inc[:k], move_idx[:k])
    if m_in > k:
        _unpack_rows(cols, inc[k:], slice(n, n_new))
    elif m_out > k:
        # h = m_out - m_in holes remain and the tail [n_new, n) has exactly h rows: every hole
        # below n_new takes one live row of the tail.  Fixed-size index algebra, no nonzero().
        holes, _ = torch.sort(move_idx[k:])
        h = holes.numel()
        in_tail = holes >= n_new
        flag = torch.zeros(h, dtype=torch.int32, device=dev)      # tail rows that are holes
        flag.index_add_(0, (holes - n_new).clamp(min=0), in_tail.to(torch.int32))
        live = flag == 0
        live_rank = torch.cumsum(live.to(torch.int64), 0) - 1
        by_rank = torch.zeros(h + 1, dtype=torch.int64, device=dev)
        by_rank[torch.where(live, live_rank, torch.full_like(live_rank, h))] = \
            torch.arange(h, device=dev) + n_new
        # the low holes are the first of the sorted list, as many as there are live tail rows
        src = torch.where(in_tail, holes, by_rank[:h])
        for c in cols:
            c[holes] = c[src]
    return n_new, (inc if m_in else None)


def exchange_rows_compact(comm, owner, pos, mom, ids, n, cap, move=None, extra=()):
    """exchange() by explicit owners (see exchange_columns).  `move` = (row numbers,
    their owners) when the leaving rows are already known (then `owner` is not needed);
    `extra`: further per-particle columns that travel with the rows.  This form finds the
    message sizes with host round trips (nonzero, bincount, all_gather); the stepping path
    gets them from the previous kick instead (ParticleStore.drift_exchange_sort).
    Returns (n_new, immigrant rows or None); the live rows are [0, n_new)."""
    P, rank = comm.world, comm.rank
    if move is None:
        move_idx = torch.nonzero(owner[:n] != rank).flatten()
        dest = owner[move_idx].long()
    else:
        move_idx, dest = move[0], move[1].long()
    send_counts = torch.bincount(dest, minlength=P).cpu().tolist()
    counts = comm.all_gather_ints(send_counts)  # counts[src][dst]
    recv_counts = counts[:, rank].tolist()
    return exchange_columns(comm, [pos, mom, ids] + list(extra), n, cap, move_idx, dest,
                            send_counts, recv_counts)


class ParticleStore:
    """The particle arrays of one rank: named per-particle columns with spare capacity, `n`
    live rows; tile sort and exchange() move all columns together.  'pos' and 'mom' are
    float64 (cap, 3) — the reference's AoS layout (species.py:2010-2064); other columns are
    float64 (cap, 3) / (cap,), int64 (cap,) or int8 (cap,)."""

    def __init__(self, domain, pos, mom, ids=None, slack=1.3, extra=None):
        """domain: a SlabDomain or a PotentialMesh (its comm may be None: one domain)"""
        self.domain = domain
        self.mesh = getattr(domain, 'mesh', domain)
        self.comm = self.mesh.comm
        self.multi = self.comm is not None and self.comm.world > 1
        n = pos.shape[0]
        cap = int(n*slack) + 1024 if self.multi else n
        dev = self.mesh.device
        self.cap, self.n = cap, n
        self.cols, self.spare = {}, {}
        self.add_column('pos', pos)
        self.add_column('mom', mom)
        # without ids the sort moves pos and mom only
        self.has_ids = ids is not None
        if ids is not None:
            self.add_column('ids', ids)
        for name, t in (extra or {}).items():
            self.add_column(name, t)
        self.table = self.mesh.new_tile_table()
        self.sorted = False
        self._expected = n
        self._slots = None
        self.last_perm = None
        # rows that the drift prepared by the last pm_kick(next_dt_over_mass=...) takes out of the
        # slab, listed by the gather-kick itself (cg_set_emigrant_list)
        P = self.comm.world if self.comm is not None else 1
        self.emig_idx = torch.empty(max(4096, cap//16), dtype=torch.int64, device=dev)
        self.emig_dest = torch.empty(self.emig_idx.numel(), dtype=torch.int32, device=dev)
        # meta: [emigrant count | send counts (P) | recv counts (P) | rows kept by the last sort]
        self.meta = torch.zeros(2 + 2*P, dtype=torch.int32, device=dev)
        self.meta_host = torch.zeros(2 + 2*P, dtype=torch.int32)
        if dev.type == 'cuda':
            self.meta_host = self.meta_host.pin_memory()
        self.meta_event = torch.cuda.Event() if dev.type == 'cuda' else None
        self._emig_for = None   # (pos pointer, n, dt_over_mass) the list was made for
        self._kept_pending = None
        self.emigrants_total = 0  # rows this rank has shipped so far (all exchanges)

    emig_count = property(lambda self: self.meta[0:1])

    # -- columns ----------------------------------------------------------------
    def add_column(self, name, values=None, dtype=None, width=None):
        dev = self.mesh.device
        if values is not None:
            shape = (self.cap,) + tuple(values.shape[1:])
            t = torch.zeros(shape, dtype=values.dtype, device=dev)
            t[:values.shape[0]] = values
        else:
            shape = (self.cap,) if width is None else (self.cap, width)
            t = torch.zeros(shape, dtype=dtype, device=dev)
        self.cols[name] = t
        self.spare.pop(name, None)

    def drop_column(self, name):
        self.cols.pop(name, None)
        self.spare.pop(name, None)

    def view(self, name):
        return self.cols[name][:self.n]

    def __getattr__(self, name):  # store.pos / .mom / .ids: the full-capacity buffers
        cols = self.__dict__.get('cols')
        if cols is not None and name in cols:
            return cols[name]
        raise AttributeError(name)

    def _others(self):
        return [k for k in self.cols if k not in ('pos', 'mom')]

    def _columns(self):
        return [self.cols['pos'], self.cols['mom']] + [self.cols[k] for k in self._others()]

    def _spare(self, name):
        t = self.spare.get(name)
        c = self.cols[name]
        if t is None or t.shape != c.shape:
            t = self.spare[name] = torch.empty_like(c)
        return t

    def resize(self, n, cap=None):
        """Set the number of live rows (new rows are zero), growing the buffers if needed."""
        cap = max(cap or 0, n)
        if cap > self.cap:
            for name, c in list(self.cols.items()):
                t = torch.zeros((cap,) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
                t[:self.n] = c[:self.n]
                self.cols[name] = t
            self.spare.clear()
            self.cap = cap
            self._slots = None
            if self.emig_idx.numel() < max(4096, cap//16):
                self.emig_idx = torch.empty(max(4096, cap//16), dtype=torch.int64,
                                            device=self.mesh.device)
                self.emig_dest = torch.empty(self.emig_idx.numel(), dtype=torch.int32,
                                             device=self.mesh.device)
        self.n = self._expected = n
        self.sorted = False
        self.touch_mom()

    def _grow(self, n_needed):
        """Room for n_needed rows: exchange() may bring in more particles than the arrays were
        sized for (the reference resizes its arrays the same way, communication.py:431-470).
        The live rows keep their places; buffers of the sort and the emigrant list follow."""
        if n_needed <= self.cap:
            return
        cap = int(n_needed*1.3) + 1024
        for name, c in list(self.cols.items()):
            t = torch.empty((cap,) + tuple(c.shape[1:]), dtype=c.dtype, device=c.device)
            t[:self.n] = c[:self.n]
            self.cols[name] = t
        self.spare.clear()
        self.cap = cap
        self._slots = None
        if self.emig_idx.numel() < max(4096, cap//16):
            self.emig_idx = torch.empty(max(4096, cap//16), dtype=torch.int64,
                                        device=self.mesh.device)
            self.emig_dest = torch.empty(self.emig_idx.numel(), dtype=torch.int32,
                                         device=self.mesh.device)
        self._emig_for = None  # (keyed on the old arrays)

    # -- bookkeeping ------------------------------------------------------------
    def check(self):
        """Raise if a kernel recorded an inconsistency since the last call (synchronises)."""
        self._check_kept(wait=True)
        self.mesh.check_errors()

    def touch_mom(self):
        """Call after changing `mom` by anything but pm_kick: the drift histogram and the
        emigrant list the last kick prepared no longer describe the particles."""
        self._emig_for = None
        self.mesh.prepare_invalidate()

    def _note_kept(self, expected):
        """Record table[-1] (rows the sort kept) for a deferred comparison: read on the next
        occasion the host waits for the GPU anyway."""
        self.meta[-1:].copy_(self.table[-1:])
        self._kept_pending = expected

    def _check_kept(self, wait):
        if self._kept_pending is None:
            return
        kept = int(self.meta[-1].item()) if wait else int(self.meta_host[-1].item())
        kept &= 0xffffffff
        expected, self._kept_pending = self._kept_pending, None
        if kept != expected:
            raise lib.ConceptGPUError(
                f'rank {self.mesh.rank}: the tile sort kept {kept} of {expected} particles — '
                'particles outside this rank\'s slab (exchange() must run after every drift)')

    # -- A11 + A12 ----------------------------------------------------------------
    def drift(self, dt_over_mass):
        self.mesh.drift(self.view('pos'), self.view('mom'), dt_over_mass)
        self.sorted = False
        self._emig_for = None

    def exchange(self):
        """exchange() (communication.py:135-517): re-home particles whose lower CIC
        cell left this rank's slab."""
        self.sorted = False
        if not self.multi:
            return
        owner = self.mesh.owner_rank(self.view('pos'))
        n_new, _ = self._exchange_owner(owner)
        self.n = self._expected = n_new

    def _exchange_owner(self, owner):
        """exchange with the message sizes found the long way (host round trips)"""
        P, rank = self.comm.world, self.comm.rank
        move_idx = torch.nonzero(owner[:self.n] != rank).flatten()
        dest = owner[move_idx].long()
        send_counts = torch.bincount(dest, minlength=P).cpu().tolist()
        recv_counts = self.comm.all_gather_ints(send_counts)[:, rank].tolist()
        self._grow(self.n - int(sum(send_counts)) + int(sum(recv_counts)))
        n_new, inc = exchange_columns(self.comm, self._columns(), self.n, self.cap, move_idx,
                                      dest, send_counts, recv_counts)
        self.emigrants_total += int(sum(send_counts))
        return n_new, inc

    def prepare_exchange(self, dt_over_mass):
        """Right after the gather-kick that listed the leavers of the coming drift: count them
        per destination on the device, swap the counts with the peers, start their copy to
        pinned memory.  Nothing here waits for the GPU."""
        m = self.mesh
        lib.check(lib.raw().cg_emigrant_dest(
            m._ctx, _vp(self.cols['pos']), _vp(self.cols['mom']), _vp(self.emig_idx),
            _vp(self.meta), self.emig_idx.numel(), float(dt_over_mass), _vp(self.emig_dest),
            _vp(self.meta[1:])))
        P = self.comm.world
        self.comm.all_to_all(self.meta[1 + P:1 + 2*P], self.meta[1:1 + P])
        self.meta_host.copy_(self.meta, non_blocking=True)
        if self.meta_event is not None:
            self.meta_event.record()
        self._emig_for = (self.cols['pos'].data_ptr(), self.n, float(dt_over_mass))

    def drift_exchange_sort(self, dt_over_mass):
        """drift + exchange + tile sort in fused form: the particles whose DRIFTED position
        leaves this rank's slab are shipped first (undrifted rows; the receiver drifts them
        with the same arithmetic), then one fused drift + sort pass pair runs over the
        compacted local set.  When the previous pm_kick prepared this drift
        (next_dt_over_mass), the leavers and every message size are already known — one wait
        for the pinned counts, then the whole step is enqueued — and the immigrants' keys are
        added to the prepared histogram so that the sort needs no histogram pass.  Same result
        as drift(); exchange(); tile_sort()."""
        m = self.mesh
        inc = None
        if self.multi:
            P = self.comm.world
            done = False
            if self._emig_for == (self.cols['pos'].data_ptr(), self.n, float(dt_over_mass)):
                if self.meta_event is not None:
                    self.meta_event.synchronize()  # the one wait of the step
                self._check_kept(wait=False)
                host = self.meta_host.tolist()
                cnt = host[0] & 0xffffffff
                if cnt <= self.emig_idx.numel():  # else the list overflowed: the long way
                    idx, dst = self.emig_idx[:cnt], self.emig_dest[:cnt]
                    self._grow(self.n - cnt + int(sum(host[1 + P:1 + 2*P])))
                    n_new, inc = exchange_columns(
                        self.comm, self._columns(), self.n, self.cap, idx, dst,
                        host[1:1 + P], host[1 + P:1 + 2*P])
                    self.emigrants_total += cnt
                    done = True
            self._emig_for = None
            if not done:
                owner = m.owner_rank_drifted(self.view('pos'), self.view('mom'), dt_over_mass)
                n_new, inc = self._exchange_owner(owner)
            self.n = n_new
        pos, mom = self.view('pos'), self.view('mom')
        if inc is not None and inc.shape[0]:
            m.prepare_rebind(pos, mom, inc[:, 0:3].contiguous(), inc[:, 3:6].contiguous())
        else:
            m.prepare_rebind(pos, mom)
        self._sort(dt_over_mass)
        self._expected = self.n
        self._note_kept(self.n)

    def tile_sort(self):
        self._sort(None)
        if not self.multi:
            return
        kept = int(self.table[-1].item()) & 0xffffffff
        if kept != self._expected:
            raise lib.ConceptGPUError(
                f'rank {self.mesh.rank}: tile sort kept {kept} of {self._expected} particles — '
                'particles outside this rank\'s slab (exchange() must run after every drift)')
        self.n = self._expected = kept

    def _sort(self, dt_over_mass):
        """(drift +) tile sort of pos and mom into the spare buffers; the other columns follow
        through the permutation (a single int64 column travels inside the sort kernel)."""
        m, n = self.mesh, self.n
        others = self._others()
        p2, m2 = self._spare('pos'), self._spare('mom')
        ride = others[0] if len(others) == 1 and self.cols[others[0]].dtype == torch.int64 \
            else None
        if ride is not None:
            i_in, i_out = self.cols[ride][:n], self._spare(ride)[:n]
        elif others:
            if self._slots is None or self._slots.numel() < self.cap:
                self._slots = torch.arange(self.cap, dtype=torch.int64, device=m.device)
                self._perm = torch.empty_like(self._slots)
            i_in, i_out = self._slots[:n], self._perm[:n]
        else:
            i_in = i_out = None
        if dt_over_mass is None:
            m.sort_particles(self.view('pos'), self.view('mom'), i_in, p2[:n], m2[:n], i_out,
                             self.table)
        else:
            m.drift_sort(self.view('pos'), self.view('mom'), i_in, p2[:n], m2[:n], i_out,
                         dt_over_mass, self.table)
        for name, new in (('pos', p2), ('mom', m2)):
            self.spare[name], self.cols[name] = self.cols[name], new
        if ride is not None:
            self.spare[ride], self.cols[ride] = self.cols[ride], self.spare[ride]
        elif others:
            # gathered straight into the spare buffer, then the buffers trade places (no copy
            # back: with the five columns of a rung run that was 0.3 ms per sort at 256^3)
            # (one pass for all of them, cg_permute_rows: a torch.index_select per column read
            # the permutation once per column — 5.7 of the 8 ms of configs[4]'s drift + sort)
            pairs = [(self.cols[name], self._spare(name)) for name in others]
            m.permute_rows(i_out, pairs)
            for name, (c, t) in zip(others, pairs):
                self.spare[name], self.cols[name] = c, t
        self.last_perm = i_out if (others and ride is None) else None
        self.sorted = True


DistributedParticles = ParticleStore  # the name bench.py and the tests grew up with


def ship_boundary_positions(mesh, pos, margin):
    """sendrecv_component (communication.py:847-1130) for x-slab domains: the positions of the
    particles within `margin` of a slab face go to the ring neighbour beyond that face.
    Returns (from_prev, from_next): the neighbours' particles near MY faces.  Only positions
    travel: the sweep is one-sided, every rank kicks its own receivers with their own rung
    factors, so neither the suppliers' rungs nor any Δmom crosses the wire."""
    comm = mesh.comm
    L, N = mesh.boxsize, mesh.gridsize
    cell = L/N
    slab_w = mesh.nxl*cell
    # lower CIC cell in [x0, x0 + nxl)  <=>  x in [xlo, xlo + slab_w); the CIC index map is
    # floor(x/cell - 0.5) on cell-centred grids and floor(x/cell) on vertex-centred ones
    # (the ownership rule of cg_owner_rank / geom_deposit)
    xlo = (mesh.x0 + 0.5*mesh.cell_centered)*cell
    rel = torch.remainder(pos[:, 0] - xlo, L)
    to_prev = pos[rel < margin]
    to_next = pos[rel >= slab_w - margin]

    def ship(send, dest, source):
        cnt = torch.tensor([send.shape[0]], dtype=torch.int64, device=send.device)
        got = torch.empty_like(cnt)
        comm.sendrecv(cnt, dest, got, source)
        recv = torch.empty((int(got.item()), 3), dtype=torch.float64, device=send.device)
        comm.sendrecv(send.contiguous(), dest, recv, source)
        return recv
    from_next = ship(to_prev, comm.prev, comm.next)   # what my next rank has near ITS lower face
    from_prev = ship(to_next, comm.next, comm.prev)
    return from_prev, from_next


def check_shortrange_fits(mesh, range_):
    cell = mesh.boxsize/mesh.gridsize
    slab_w = mesh.nxl*cell
    P = mesh.nprocs
    if P > 1 and (range_*1.001 >= slab_w or (P == 2 and 2.002*range_ >= slab_w)):
        raise lib.ConceptGPUError(
            f'short-range force range {range_:g} too large for slabs of width {slab_w:g} '
            f'({P} domains): boundary particles would be needed from beyond the ring neighbours')


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
    m = domain.mesh
    L = m.boxsize
    nt = int((L/1)/tilesize*(1 + commons.machine_ϵ))  # global tiling, species.py:3943-3950
    if nt < 4:
        raise lib.ConceptGPUError(
            'The global gravity tiling needs to have at least 4 tiles across the box in every '
            'direction (species.py:3971)')
    check_shortrange_fits(m, range_)
    pos = particles.view('pos')
    n = pos.shape[0]
    if m.nprocs == 1:
        ghosts = []
    else:
        ghosts = list(ship_boundary_positions(m, pos, range_*(1 + 1e-9) + 1e-9*L))
    supp = torch.cat([pos] + ghosts).contiguous()
    ext = L/nt
    cells_r = m.shortrange_cells(pos.contiguous(), nt, ext)
    cells_s = m.shortrange_cells(supp, nt, ext) if ghosts else cells_r
    table, maxr2 = shortrange.get_shortrange_table(softening, scale, range_, tablesize, kernel,
                                                   pos.device)
    dmom = torch.zeros((n, 3), dtype=torch.float64, device=pos.device)
    m.shortrange_sweep_cells(cells_r, dmom, cells_s, nt, table, (tablesize - 1)/maxr2, range_**2,
                             factor)
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
    fold = m.fold_ghosts_start()   # the ghost layer travels under the first transforms
    mark('deposit')
    m.poisson_solve(deconv_order, C, long_range, E, fold_finish=fold, fill=True)
    mark('poisson+transposes+halos')
    if next_dt_over_mass is None:
        particles._emig_for = None
        m.gather_kick_tiled(particles.view('pos'), particles.view('mom'), particles.table,
                            diff_order, kick_factor)
    else:
        # also histogram the tile keys after the NEXT drift and list the rows it takes out of the
        # slab (for drift_exchange_sort)
        multi = particles.multi
        if multi:
            m.set_emigrant_list(particles.emig_idx, particles.emig_count)
        try:
            m.gather_kick_tiled_prepare(particles.view('pos'), particles.view('mom'),
                                        particles.table, diff_order, kick_factor,
                                        next_dt_over_mass)
        finally:
            if multi:  # the launch holds the pointers; the context must not
                m.set_emigrant_list(None, None)
        if multi:
            particles.prepare_exchange(next_dt_over_mass)
    mark('gather_kick')


class RegionParticles:
    """The streaming form of the particle arrays for the PM step: pos / mom (/ ids) kept in
    tile REGIONS WITH GAPS (include/concept_gpu.h, cg_gather_kick_drift_scatter), two buffer
    sets in ping-pong.  A step is

        deposit() -> mesh.poisson_solve(...) -> kick_drift_sort(...)

    — the long-range kick, the drift that follows it in the time loop (main.py:335-358) and
    the tile sort of the drifted particles in one pass.  On x-slab domains the pass also hands
    over the particles leaving the slab; finish_exchange() (called by the next deposit()) ships
    them and seats the arrivals, after the step's single wait for the GPU (the message sizes).
    Built from a tile-sorted ParticleStore; dense() converts back."""

    def __init__(self, store, slack=1.25, drop_order=False, drop_ids=False):
        """drop_order: the `order` column is known to equal `ids` (a Component whose
        identifiers are its running row numbers): one 64-bit column travels instead of two
        (8.6 GB less traffic per pass at 2^28 particles), columns() returns no 'order'.
        drop_ids: no 64-bit column travels at all — the reference's particles without
        identifiers (species.py:2040-2064: `ids` only where the run uses them), whose memory
        order the time loop is free to change; columns() returns neither."""
        if not store.sorted:
            raise lib.ConceptGPUError('RegionParticles: the store must be tile-sorted')
        m = self.mesh = store.mesh
        self.comm, self.multi = store.comm, store.multi
        dev = m.device
        n = store.n
        self.n_hint = n
        cap = m.region_capacity(int(n*slack) + 1024 if self.multi else n)
        self.cap = cap
        # two 64-bit columns may travel with the particles: `ids` and `order` (a Component's
        # identifiers and the row numbers its host() restores the populated order with)
        self.has_ids = 'ids' in store.cols and not drop_ids
        self.has_aux = 'order' in store.cols and not (drop_order and self.has_ids) \
            and not drop_ids
        mk = lambda w: torch.empty((cap, w), dtype=torch.float64, device=dev)
        mi = lambda have: [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(2)] \
            if have else [None, None]
        self.pos, self.mom = [mk(3), mk(3)], [mk(3), mk(3)]
        self.ids, self.aux = mi(self.has_ids), mi(self.has_aux)
        self.pos[0][:n] = store.view('pos')
        self.mom[0][:n] = store.view('mom')
        if self.has_ids:
            self.ids[0][:n] = store.view('ids')
        if self.has_aux:
            self.aux[0][:n] = store.view('order')
        self.cur = 0
        # the present order: dense (the store's tile table) until the first fused pass
        self.start = self._dense_table = store.table[:8*m.ntiles + 1].clone()
        self.count = None
        self.tables = [m.new_region_table(), m.new_region_table()]
        self.n_dense = n
        # leavers of the fused pass: rows of 8 doubles, their destinations, the counts
        P = self.comm.world if self.comm is not None else 1
        rows_cap = max(4096, cap//16)
        self.rows = torch.empty((rows_cap, 8), dtype=torch.float64, device=dev)
        self.rows_dest = torch.empty(rows_cap, dtype=torch.int32, device=dev)
        self.meta = torch.zeros(1 + 2*P, dtype=torch.int32, device=dev)
        self.meta_host = torch.zeros(1 + 2*P, dtype=torch.int32)
        if dev.type == 'cuda':
            self.meta_host = self.meta_host.pin_memory()
        self.meta_event = torch.cuda.Event()
        self.pending = False
        self.emigrants_total = 0
        # sum of |mom|^2 left by the last fused pass (cg_set_momentum_sum): what the time loop's
        # v_rms needs after every kick, without a pass over the momenta of its own
        self.m2 = torch.zeros(1, dtype=torch.float64, device=dev)
        self.m2_host = torch.zeros(1, dtype=torch.float64)
        if dev.type == 'cuda':
            self.m2_host = self.m2_host.pin_memory()
        self.m2_event = torch.cuda.Event()
        self.m2_valid = False

    # -- A1 -------------------------------------------------------------------------
    def deposit(self, contribution, accumulate=False):
        self.finish_exchange()
        m, c = self.mesh, self.cur
        if self.count is None:
            m.deposit_tiled(self.pos[c][:self.n_dense], self._dense_table, contribution,
                            accumulate)
        else:
            m.deposit_regions(self.pos[c], self.start, self.count, contribution, accumulate)

    # -- A9/A10 + A11 + tile sort (+ the hand-over of A12) ----------------------------
    def kick_drift_sort(self, diff_order, kick_factor, dt_over_mass):
        m, c, o = self.mesh, self.cur, 1 - self.cur
        start_out, count_out = self.tables[o]
        m.predict_regions(self.start, self.count, start_out)
        if self.multi:
            lib.check(lib.raw().cg_set_emigrant_rows(m._ctx, _vp(self.rows), _vp(self.meta),
                                                     self.rows.shape[0]))
        lib.check(lib.raw().cg_set_momentum_sum(m._ctx, _vp(self.m2)))
        try:
            m.gather_kick_drift_scatter(self.pos[c], self.mom[c], self.ids[c], self.start, self.count,
                                        self.pos[o], self.mom[o], self.ids[o], start_out,
                                        count_out, diff_order, kick_factor, dt_over_mass,
                                        aux_in=self.aux[c], aux_out=self.aux[o])
        finally:
            lib.check(lib.raw().cg_set_momentum_sum(m._ctx, None))
            if self.multi:
                lib.check(lib.raw().cg_set_emigrant_rows(m._ctx, None, None, 0))
        self.m2_host.copy_(self.m2, non_blocking=True)
        self.m2_event.record()
        self.m2_valid = True
        self.start, self.count, self.cur = start_out, count_out, o
        if self.multi:
            # destinations and counts on the device, counts swapped with the peers, copy to
            # pinned memory started: nothing here waits for the GPU
            P = self.comm.world
            lib.check(lib.raw().cg_emigrant_rows_dest(
                m._ctx, _vp(self.rows), _vp(self.meta), self.rows.shape[0], _vp(self.rows_dest),
                _vp(self.meta[1:])))
            self.comm.all_to_all(self.meta[1 + P:1 + 2*P], self.meta[1:1 + P])
            self.meta_host.copy_(self.meta, non_blocking=True)
            self.meta_event.record()
            self.pending = True

    def measure_momentum(self, want_max=True):
        """(Σ mom², max |mom_i|²) of this rank's live particles (analysis.measure's inputs).
        want_max=False: the sum the last fused pass left — over the particles THIS rank
        kicked, the leavers of its slab included, so that the ranks' sums add up to the
        box's — with NaN for the maximum: no pass, no kernel."""
        self.finish_exchange()
        if not want_max and self.m2_valid:
            self.m2_event.synchronize()
            return float(self.m2_host[0]), float('nan')
        if self.count is None:
            return self.mesh.measure_momentum(self.mom[self.cur][:self.n_dense])
        return self.mesh.measure_momentum_regions(self.mom[self.cur], self.start, self.count)

    def finish_exchange(self):
        """ship the leavers of the last kick_drift_sort and seat the arrivals (x-slab
        domains; the one wait for the GPU of a step)"""
        if not self.pending:
            return
        self.pending = False
        m, P = self.mesh, self.comm.world
        self.meta_event.synchronize()
        host = self.meta_host.tolist()
        # (more leavers than the row buffer holds: the pass flagged CG_ERR_BUCKET_OVERFLOW and
        # the counts below are those of the rows that are there — every rank goes through the
        # same exchange, the caller then undoes the pass: stepper.timeloop / check())
        cnt = min(host[0] & 0xffffffff, self.rows.shape[0])
        send_counts, recv_counts = host[1:1 + P], host[1 + P:1 + 2*P]
        m_in = int(sum(recv_counts))
        order = torch.argsort(self.rows_dest[:cnt].long(), stable=True)
        rows = self.rows[:cnt][order].contiguous()
        inc = torch.empty((m_in, 8), dtype=torch.float64, device=m.device)
        self.comm.all_to_all(inc, rows, recv_counts, send_counts)
        self.emigrants_total += cnt
        if m_in:
            c = self.cur
            lib.check(lib.raw().cg_region_insert(
                m._ctx, _vp(inc), m_in, _vp(self.start), _vp(self.count), _vp(self.pos[c]),
                _vp(self.mom[c]), _vp(self.ids[c]) if self.has_ids else None,
                _vp(self.aux[c]) if self.has_aux else None, self.cap))

    def snapshot(self):
        """The present order (which buffer set, its region tables): kick_drift_sort writes the
        other set only, so restore(snapshot) undoes a pass whose regions overflowed."""
        self.finish_exchange()
        return (self.cur, self.start, self.count)

    def restore(self, snap):
        self.cur, self.start, self.count = snap
        self.pending = False  # (the leavers of the undone pass are dropped with it)
        self.m2_valid = False

    # -- bookkeeping ------------------------------------------------------------------
    @property
    def n(self):
        """live particles of this rank (synchronises)"""
        self.finish_exchange()
        if self.count is None:
            return self.n_dense
        return int(self.count.long().sum().item())

    def check(self):
        self.finish_exchange()
        self.mesh.check_errors()

    def columns(self):
        """{'pos', 'mom' (, 'ids', 'order')} of the live particles as dense tensors, in tile
        order"""
        self.finish_exchange()
        c = self.cur
        if self.count is None:
            sel = slice(0, self.n_dense)
        else:
            st, ct = self.start.long(), self.count.long()
            slot = torch.arange(self.cap, device=self.mesh.device)
            k = (torch.searchsorted(st, slot, right=True) - 1).clamp(max=ct.numel() - 1)
            sel = (slot - st[k]) < ct[k]
        out = {'pos': self.pos[c][sel], 'mom': self.mom[c][sel]}
        if self.has_ids:
            out['ids'] = self.ids[c][sel]
        if self.has_aux:
            out['order'] = self.aux[c][sel]
        return out

    def dense(self):
        """(pos, mom, ids or None) of the live particles as dense tensors, in tile order"""
        cols = self.columns()
        return cols['pos'], cols['mom'], cols.get('ids')


def pm_step_regions(domain, rp, contribution, deconv_order, C, kick_factor, dt_over_mass,
                    diff_order=2, long_range=False, E=0.0, mark=None):
    """One PM step of particles kept in regions (RegionParticles), sharded: exchange of the
    last step's leavers + deposit, the Poisson solve with its transposes and halos, then kick,
    drift and tile sort in one pass."""
    mark = mark or (lambda name: None)
    m = domain.mesh
    rp.deposit(contribution)
    fold = m.fold_ghosts_start()
    mark('exchange+deposit')
    m.poisson_solve(deconv_order, C, long_range, E, fold_finish=fold, fill=True)
    mark('poisson+transposes+halos')
    rp.kick_drift_sort(diff_order, kick_factor, dt_over_mass)
    mark('kick_drift_sort')
