"""concept_amd.comm — the process group of the x-slab domain decomposition.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The
reference's counterpart is the module-level MPI state of commons.py:174-260
(`comm`, `rank`, `nprocs`, `master`) that every collective in communication.py
uses.  `init()` makes a group the ACTIVE decomposition: meshes created through
`mesh.get_mesh()` and `species.Component`s then live on x-slab domains, and
`interactions.gravity()` is collective over the group, as the reference's is
over MPI ranks (interactions.py:2854-2961).

The same code runs under "gloo" (tests: ranks sharing one GPU, or CPU-only
checks of the exchange logic) by staging messages through host memory.
"""
import os

import torch
import torch.distributed as dist

_active = None


class Comm:
    """Thin wrapper over torch.distributed that also works on gloo."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.stage = self.backend != 'nccl'  # gloo: stage device tensors through the host
        self.next = (self.rank + 1) % self.world
        self.prev = (self.rank - 1) % self.world
        self.last_sent = 0
        # CONCEPT_GPU_COMM_SELF=1: messages of a rank to itself go through the transport as
        # well (a one-rank RCCL group then exercises every send / receive call of the sharded
        # path on the one GPU there is: tests)
        self.self_transport = os.environ.get('CONCEPT_GPU_COMM_SELF') == '1'
        # CONCEPT_GPU_DRY_LINKS=<GB/s>: the FFT transposes move no data — each is replaced by a
        # device sleep of (bytes one rank sends to ONE peer) / rate on a side stream, the way
        # RCCL would occupy one xGMI link per peer (bench.py --dry-links: the pipelining
        # schedule of the transposing solve is exercised, and its overlap with the transforms
        # measured, where there is no second GPU).  The results are garbage: a timing mode.
        self.dry_rate = float(os.environ.get('CONCEPT_GPU_DRY_LINKS', '0') or 0)*1e9
        self.dry_ms = 0.0          # sleep time requested so far
        self._dry_stream = None
        self._dry_cycles_per_ms = None

    def _dry(self, bytes_per_peer):
        """a sleep of bytes_per_peer / dry_rate on the side stream, behind what the current
        stream has queued; returns a work-like object whose wait() joins it"""
        if self._dry_stream is None:
            self._dry_stream = torch.cuda.Stream()
            # calibrate torch.cuda._sleep (cycles of the device's timer)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(1000)
            torch.cuda.synchronize()
            e0.record()
            torch.cuda._sleep(20_000_000)
            e1.record()
            torch.cuda.synchronize()
            self._dry_cycles_per_ms = 20_000_000/max(e0.elapsed_time(e1), 1e-3)
        ms = bytes_per_peer/self.dry_rate*1e3
        self.dry_ms += ms
        side = self._dry_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(ms*self._dry_cycles_per_ms))
            ev = torch.cuda.Event()
            ev.record(side)

        class _Work:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(ev)
        return _Work()

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self.dry_rate and out_splits is None and inp.numel()*inp.element_size() >= 1 << 20:
            # (a transpose buffer: the small all-to-alls of counts and particle rows are real)
            self._dry(inp.numel()*inp.element_size()/self.world).wait()
            return
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
        if self.dry_rate:
            w = self._dry(i[0].numel()*i.element_size())
            if async_op:
                return w
            w.wait()
            return None
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
        if dest == self.rank and source == self.rank and not self.self_transport:
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

    def sendrecv_start(self, send, dest, recv, source):
        """sendrecv posted without waiting: returns finish(), which makes the current stream
        wait for the message (RCCL runs it on its own stream behind what the current stream
        has queued so far, so kernels launched in between overlap it).  The buffers must stay
        untouched until finish().  gloo (tests): completed here, finish() does nothing."""
        if (dest == self.rank and source == self.rank and not self.self_transport) or self.stage:
            self.sendrecv(send, dest, recv, source)
            return lambda: None
        ops = []
        if send.numel():
            ops.append(dist.P2POp(dist.isend, send, dest, group=self.group))
        if recv.numel():
            ops.append(dist.P2POp(dist.irecv, recv, source, group=self.group))
        works = dist.batch_isend_irecv(ops) if ops else []

        def finish():
            for w in works:
                w.wait()
        return finish

    def all_gather_ints(self, values):
        t = torch.tensor(values, dtype=torch.int64)
        if not self.stage:
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.stack(out).cpu()

    def all_gather_floats(self, values):
        """float64 [world, len(values)] of host numbers, in rank order (every rank gets the
        same array: a sum over it is the same on every rank, bit for bit)"""
        t = torch.tensor(values, dtype=torch.float64)
        if not self.stage:
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return torch.stack(out).cpu()

    def all_gather_rows(self, t):
        """Concatenation over the ranks (in rank order) of tensors that differ in their first
        dimension only; returned on t's device."""
        counts = self.all_gather_ints([t.shape[0]])[:, 0].tolist()
        m = max(counts) if counts else 0
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype)
        pad[:t.shape[0]] = t.cpu() if self.stage else t.cpu()
        if not self.stage:
            pad = pad.to(t.device)
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        return torch.cat([o[:c] for o, c in zip(out, counts)]).to(t.device)

    def any(self, flag):
        """Logical OR of a host bool over the ranks."""
        return bool(self.all_gather_ints([int(bool(flag))]).sum().item())


def init(group=None, force=False):
    """Make `group` (default: the world group) the active domain decomposition.  A group of
    one rank is ignored unless force (tests of the transposing path on one GPU)."""
    global _active
    c = Comm(group)
    c.force = bool(force) or os.environ.get('CONCEPT_GPU_DIST_FORCE') == '1'
    _active = c if (c.world > 1 or c.force) else None
    from . import mesh
    mesh.free_meshes()  # meshes of another decomposition must not be reused
    return _active


def shutdown():
    global _active
    _active = None
    from . import mesh
    mesh.free_meshes()


def active():
    return _active
