"""concept_amd.snapshot — initial-condition ingestion: GADGET-2 snapshots
(SnapFormat 1 and 2), the format the reference writes and reads with its GadgetSnapshot
class (snapshot.py:640-2640), so that the GPU stepper can start from particle data the
reference (or GADGET / N-GenIC) produced elsewhere.  SURVEY.md §8(f) row 4.

What is restated: the block structure (read_block_bgn, snapshot.py:2390-2445), the 256-byte
HEAD block (header_fields, snapshot.py:658-689), the POS / VEL / ID blocks with 32- or
64-bit payloads and their unit conversion (get_blocks_info, snapshot.py:1520-1553):
    pos = POS * unit_length                       (wrapped into [0, boxsize))
    mom = VEL * unit_velocity * mass * Time**1.5  (GADGET stores u = a dx/dt / sqrt(a))
    mass = Massarr[type] * unit_mass
with the default GADGET units 'kpc/h', 'km/s', '10¹⁰ m☉/h' (commons.py:2787-2804).
Snapshots of one file or split over several (<base>.0, <base>.1, ...: named by their directory,
their first file or a pattern, snapshot.py:1821-1925); per-type masses in the header (a MASS
block with individual masses aborts by name).
Host-side I/O: numpy only; `to_components()` uploads to GPU Components.

Over several domains (comm.init) every rank reads ITS SHARE of every component — the rows
[start_local, start_local + N_local) of communication.partition() (communication.py:39-56), the
byte ranges of the POS / VEL / ID blocks that hold them, as the reference's loader does
(snapshot.py:2066-2330) — and `to_components()` hands them to Component.populate_local(), whose
exchange() re-homes them to the slabs that own them: no rank ever holds the whole file.

`save()` writes the same format (the writer of snapshot.py:880-1512, one file or several): HEAD block as
`populate()` builds it (snapshot.py:2462-2560), then POS / VEL / ID with the reference's
conversion — value*(1/unit) (the reciprocal is formed first, snapshot.py:1375-1392), positions
that reach the box size in file units after the conversion wrapped by one box, the result cast
to the block's width; components without identifiers get running numbers that continue over the
components (snapshot.py:1343-1349, 1406-1412).  Files written here are byte-identical to those
the reference writes from the same arrays (tests/test_snapshot.py, both fixtures)."""
import os
import struct

import numpy as np

from . import commons
from .lib import ConceptGPUError

component_names = [f'GADGET {t}' for t in ('gas', 'halo', 'disk', 'bulge', 'stars', 'bndry')]
num_particle_types = len(component_names)
# name, struct format (snapshot.py:658-689); the rest of the 256 bytes is padding
header_fields = (
    ('Npart', '6I'), ('Massarr', '6d'), ('Time', 'd'), ('Redshift', 'd'), ('FlagSfr', 'i'),
    ('FlagFeedback', 'i'), ('Nall', '6I'), ('FlagCooling', 'i'), ('NumFiles', 'i'),
    ('BoxSize', 'd'), ('Omega0', 'd'), ('OmegaLambda', 'd'), ('HubbleParam', 'd'),
    ('FlagAge', 'i'), ('FlagMetals', 'i'), ('NallHW', '6I'), ('flag_entr_ics', 'i'))
headersize = 256
default_units = {'length': 'kpc/h', 'velocity': 'km/s', 'mass': '10**10*m_sun/h'}


def partition(size, rank, nprocs):
    """communication.partition (communication.py:39-56): (start_local, size_local) of a fair
    split of `size` rows; the higher ranks take the extra rows."""
    size_local = size//nprocs
    start_local = rank*size_local
    rank_transition = nprocs + size_local*nprocs - size
    if rank >= rank_transition:
        size_local += 1
        start_local += rank - rank_transition
    return start_local, size_local


def _unit(expr, h, p):
    ns = dict(vars(p.units))
    ns['h'] = h
    return float(eval(expr, {}, ns))


def correct_float(val_raw):
    """commons.correct_float (commons.py:5356-5388), which the reference's writer applies to
    every double of the header (snapshot.py:1731-1733): the float with the shortest decimal
    form within ±10 ε of the value, when that form is shorter by more than two characters
    (33599.99999999999 → 33600.0)."""
    val_raw = float(val_raw)
    val_g = float(f'{val_raw:g}')
    if val_g == val_raw:
        return val_g
    val_str = str(abs(val_raw))
    if 'e' in val_str:
        val_str = val_str[:val_str.index('e')]
    if len(val_str.replace('.', '')) < 15:
        return val_raw
    eps = float(np.finfo(np.float64).eps)
    val_new = val_raw*(1 - 10*eps)
    upper = val_raw*(1 + 10*eps)
    val_correct = val_new
    while val_new <= upper:
        if len(str(val_new)) < len(str(val_correct)):
            val_correct = val_new
        val_new = float(np.nextafter(val_new, np.inf))
    return val_correct if len(str(val_correct)) < len(str(val_raw)) - 2 else val_raw


def divvy(num_particles, file_max):
    """GadgetSnapshot.divvy (snapshot.py:1424-1512): how many particles of each component go to
    each file of a snapshot whose files hold at most `file_max` particles — the number of files
    from filling them to the brim, then the particles spread evenly over that many files, the
    fuller files (and those with more particles of the lower types) first."""
    num_particles = [int(n) for n in num_particles]
    tot = sum(num_particles)

    def fill(file_max):
        num_files = tot//file_max + 1
        common = [max(n//num_files, 1) for n in num_particles]
        remaining = list(num_particles)
        files = []
        while sum(remaining) > 0:
            row = []
            for j, c in enumerate(common):
                c = min(c, remaining[j])
                row.append(c)
                remaining[j] -= c
            files.append(row)
        i_left, i_right = 0, len(files) - 1
        while i_left != i_right:
            left = files[i_left]
            n_left = sum(left)
            if n_left == file_max:
                i_left += 1
                continue
            right = files[i_right]
            for j, move in enumerate(list(right)):
                if move + n_left > file_max:
                    move = file_max - n_left
                right[j] -= move
                left[j] += move
                n_left += move
            if sum(right) == 0:
                i_right -= 1
        for i in range(len(files) - 1, 0, -1):
            if sum(files[i]) > 0:
                break
            files.pop()
        return files
    num_files = len(fill(file_max))
    even = tot//num_files
    even += (even*num_files < tot)
    files = fill(even)
    files.sort(key=lambda row: (sum(row), row), reverse=True)
    if len(files) != num_files or sum(map(sum, files)) != tot or max(map(sum, files)) > file_max:
        raise ConceptGPUError('Something went wrong divvying up the particles')
    return files


class GadgetSnapshot:
    """load(filename) fills .header, .params and .components — dicts with name, species,
    N, mass and float64 arrays pos (N, 3), mom (N, 3) and ids (N,) or None, in the unit
    system of commons.params."""
    name = 'GADGET'

    def __init__(self, params=None, units=None, rank=None, nprocs=None):
        """rank / nprocs: read only this rank's share of every component (default: the active
        domain decomposition's, concept_amd.comm; one domain: everything)"""
        if rank is None:
            from . import comm
            active = comm.active()
            rank, nprocs = (active.rank, active.world) if active is not None else (0, 1)
        self.rank, self.nprocs = int(rank), int(nprocs)
        self.p = params or commons.params
        if self.p is None:
            raise ConceptGPUError('no parameters loaded: call concept_amd.commons.load_params()')
        self.unit_expr = dict(default_units)
        self.unit_expr.update(units or {})
        self.header, self.params, self.components = {}, {}, []
        self.snapformat = None

    # -- low level -----------------------------------------------------------
    @staticmethod
    def get_snapformat(f):
        """SnapFormat 2 files open with a 8-byte record holding the block name"""
        f.seek(0)
        first = f.read(4)
        if len(first) < 4:
            return -1
        size = struct.unpack('<I', first)[0]
        if size == 8:
            return 2
        if size == headersize:
            return 1
        return -1

    def _block(self, f, offset):
        """-> (payload offset, payload size, name or '', offset of the next block)"""
        f.seek(offset)
        name = ''
        if self.snapformat == 2:
            rec = f.read(16)
            if len(rec) < 16:
                return None
            s0, nm, nxt, s1 = struct.unpack('<I4sII', rec)
            if s0 != 8 or s1 != 8:
                raise ConceptGPUError(f'{self.filename}: malformed block name record at {offset}')
            name = nm.decode('utf8').rstrip()
            offset += 16
            f.seek(offset)
        head = f.read(4)
        if len(head) < 4:
            return None
        size = struct.unpack('<I', head)[0]
        payload = offset + 4
        f.seek(payload + size)
        tail = f.read(4)
        if len(tail) < 4 or struct.unpack('<I', tail)[0] != size:
            raise ConceptGPUError(f'{self.filename}: block "{name}" at {offset} is not framed by '
                                  f'its size ({size})')
        if self.snapformat == 2 and nxt != size + 8:
            raise ConceptGPUError(f'{self.filename}: size of block "{name}" not consistent: '
                                  f'{nxt} - 8 ≠ {size}')
        return payload, size, name, payload + size + 4

    def read_header(self, f):
        blk = self._block(f, 0)
        if blk is None:
            raise ConceptGPUError('Expected block "HEAD" at the beginning of the file but found '
                                  'nothing')
        payload, size, name, nxt = blk
        if self.snapformat == 2 and name != 'HEAD':
            raise ConceptGPUError(f'Expected block "HEAD" at the beginning of the file but found '
                                  f'"{name}"')
        if size != headersize:
            raise ConceptGPUError(f'Block "HEAD" has size {size} but expected {headersize}')
        f.seek(payload)
        header = {}
        for key, fmt in header_fields:
            t = struct.unpack('<' + fmt, f.read(struct.calcsize('<' + fmt)))
            header[key] = t[0] if len(t) == 1 else list(t)
        return header, nxt

    # -- load ------------------------------------------------------------------
    def _files(self, filename):
        """The files of the snapshot `filename` names (snapshot.py:1821-1858): the file itself;
        or, for a directory, a name ending in '.0' or in '*', the sequence <base>.0, <base>.1,
        ... as far as it exists."""
        def is_gadget(fn):
            try:
                with open(fn, 'rb') as f:
                    return self.get_snapformat(f) in (1, 2)
            except OSError:
                return False
        if os.path.isfile(filename) and not filename.endswith(('*', '.0')):
            return [filename]
        if os.path.isdir(filename):
            import glob
            firsts = [fn for fn in sorted(glob.glob(f'{filename}/*.0')) if is_gadget(fn)]
            if len(firsts) != 1:
                msg = ', '.join(f'"{fn}"' for fn in firsts)
                raise ConceptGPUError(f'Found several candidates for the first {self.name} '
                                      f'snapshot file: {msg}' if firsts else
                                      f'Could not locate {self.name} snapshot "{filename}"')
            filename = firsts[0]
        base = filename[:-2] if filename.endswith('.0') else filename
        base = base.rstrip('.*')
        files = []
        while os.path.isfile(f'{base}.{len(files)}') and is_gadget(f'{base}.{len(files)}'):
            files.append(f'{base}.{len(files)}')
        if not files:
            raise ConceptGPUError(f'Could not locate {self.name} snapshot "{filename}"')
        return files

    def load(self, filename, only_params=False):
        files = self._files(filename)
        self.filename = filename = files[0]
        p = self.p
        # the header of every file (snapshot.py:1880-1925): the first one counts, the others
        # contribute their Npart
        npart_files, offsets = [], []
        for i, fn in enumerate(files):
            with open(fn, 'rb') as f:
                if i == 0:
                    self.snapformat = self.get_snapformat(f)
                    if self.snapformat not in (1, 2):
                        raise ConceptGPUError(
                            f'Could not determine GADGET SnapFormat of "{filename}"')
                header_i, offset_i = self.read_header(f)
            if i == 0:
                header = self.header = header_i
                num_files = max(int(header['NumFiles']), 1)
                if num_files > len(files):
                    msg = (f'Could only locate {len(files)} of the supposed {num_files} files '
                           'making up the snapshot.')
                    if not filename.endswith('.0'):
                        msg += f' Is "{filename}" not the first file of the snapshot?'
                    raise ConceptGPUError(msg)
                files = files[:num_files]   # (more files than the header counts: ignored)
            elif i >= len(files):
                break
            npart_files.append([int(n) for n in header_i['Npart']])
            offsets.append(offset_i)
        npart_files = npart_files[:len(files)]
        h = header['HubbleParam']
        if h == 0:
            raise ConceptGPUError(f'{filename}: HubbleParam = 0 in the header')
        self.h = h
        self.unit_length = _unit(self.unit_expr['length'], h, p)
        self.unit_velocity = _unit(self.unit_expr['velocity'], h, p)
        self.unit_mass = _unit(self.unit_expr['mass'], h, p)
        u = p.units
        self.params = {'H0': h*(100*u.km/(u.s*u.Mpc)), 'a': header['Time'],
                       'boxsize': header['BoxSize']*self.unit_length,
                       'Ωm': header['Omega0'], 'ΩΛ': header['OmegaLambda']}
        npart = [sum(nf[j] for nf in npart_files) for j in range(num_particle_types)]
        # Nall + 2³² NallHW must agree with the files' Npart — in the standard convention or in
        # N-GenIC's, which keeps the high word of type 1 in Nall[2] (snapshot.py:1931-1952)
        nall, nhw = list(header['Nall']), list(header['NallHW'])
        alt_nall, alt_nhw = list(nall), list(nhw)
        alt_nhw[1], alt_nall[2] = alt_nall[2], 0
        if not any(all(a_ + 2**32*w_ in (n, 0) for a_, w_, n in zip(A, W, npart))
                   for A, W in ((nall, nhw), (alt_nall, alt_nhw))):
            raise ConceptGPUError(
                f'{filename}: inconsistent particle counts in the header: Nall = {nall}, '
                f'NallHW = {nhw}, while Npart summed over the files is {npart}')
        self.components = []
        for j, n in enumerate(npart):
            if n == 0:
                continue
            mass = header['Massarr'][j]
            if mass <= 0:
                raise ConceptGPUError(
                    f'Mass of "{component_names[j]}" particles is {mass}×10¹⁰ h⁻¹ m☉ '
                    '(individual particle masses, block MASS, are not read)')
            start_local, n_local = partition(n, self.rank, self.nprocs)
            self.components.append({'name': component_names[j], 'species': 'matter', 'N': n,
                                    'mass': mass*self.unit_mass, 'pos': None, 'mom': None,
                                    'ids': None, 'start_local': start_local,
                                    'N_local': n_local, 'type': j})
        if only_params:
            return self
        boxsize = self.params['boxsize']
        raw = {}   # (component index, block) -> this rank's rows, in file order
        done = [0]*num_particle_types   # rows of each type in the files before this one
        for fn, npf, offset in zip(files, npart_files, offsets):
            ntot = sum(npf)
            with open(fn, 'rb') as f:
                order = ['POS', 'VEL', 'ID']  # SnapFormat 1: blocks are identified by position
                seen = 0
                while ntot:
                    blk = self._block(f, offset)
                    if blk is None:
                        break
                    payload, size, name, offset = blk
                    if self.snapformat == 1:
                        if seen >= len(order):
                            break
                        name = order[seen]
                    seen += 1
                    if name not in ('POS', 'VEL', 'ID'):
                        continue  # Skipping block (e.g. MASS, U)
                    if size % ntot:
                        raise ConceptGPUError(
                            f'File {fn} contains {ntot} particles but its "{name}" block has '
                            f'a size of {size} bytes, which does not divide the particle number.')
                    per = size//ntot
                    if name == 'ID':
                        if per not in (4, 8):
                            raise ConceptGPUError(f'ID block with {per} bytes per particle')
                        dtype, width = ('<u4' if per == 4 else '<u8'), 1
                    else:
                        if per not in (12, 24):
                            raise ConceptGPUError(f'No data format with a size of {per//3} bytes '
                                                  f'implemented for block "{name}"')
                        dtype, width = ('<f4' if per == 12 else '<f8'), 3
                    for ci, c in enumerate(self.components):
                        j = c['type']
                        # this file holds rows [g0, g1) of the component; mine are [lo, hi)
                        g0, g1 = done[j], done[j] + npf[j]
                        lo, hi = max(g0, c['start_local']), min(g1, c['start_local'] + c['N_local'])
                        if hi <= lo:
                            continue
                        first = sum(npf[:j]) + (lo - g0)   # row of the block
                        f.seek(payload + first*per)
                        part = np.fromfile(f, dtype=dtype, count=width*(hi - lo))
                        raw.setdefault((ci, name), []).append(part)
            for j in range(num_particle_types):
                done[j] += npf[j]
        for ci, c in enumerate(self.components):
            for name, key in (('POS', 'pos'), ('VEL', 'mom'), ('ID', 'ids')):
                parts = raw.get((ci, name))
                if parts is None:
                    if c['N_local'] == 0 and name != 'ID':
                        c[key] = np.zeros((0, 3))
                        continue
                    if name == 'ID':
                        continue
                    raise ConceptGPUError(f'Could not find required block "{name}"')
                data = np.concatenate(parts) if len(parts) > 1 else parts[0]
                if name == 'ID':
                    c['ids'] = data.astype(np.int64)
                    continue
                part = data.astype(np.float64).reshape(c['N_local'], 3)
                if name == 'POS':
                    pos = part*self.unit_length
                    pos[pos >= boxsize] -= boxsize  # round-off safeguard
                    c['pos'] = np.ascontiguousarray(pos)
                else:
                    unit = self.unit_velocity*c['mass']*header['Time']**1.5
                    c['mom'] = np.ascontiguousarray(part*unit)
        return self

    # -- save ------------------------------------------------------------------
    def save(self, components, filename, a=None, params=None, snapformat=2, dataformat=None,
             header=None, particles_per_file='automatic', output_base='snapshot'):
        """One-file GADGET snapshot of particle components.  components: Components (or the
        dicts load() makes) whose names are GADGET type names ('GADGET halo', ...); a single
        matter / cold dark matter component of another name is written as the halo type
        (snapshot.py:2470-2505).  a: scale factor (header Time); params: overrides for H0,
        boxsize, Ωm, ΩΛ (snapshot.py:2517-2527); dataformat: bits of 'POS', 'VEL' (32 | 64) and
        'ID' (32 | 64 | 'automatic'); header: fields to overwrite (gadget_snapshot_params
        ['header']).  On several domains the particles are gathered (Component.host) and rank 0
        writes.  particles_per_file: gadget_snapshot_params['particles per file'] — 'automatic':
        as many as the format's 32-bit block size holds (snapshot.py:706-732); a snapshot that
        needs several files becomes the directory `filename` with snapshot.0, snapshot.1, ...
        (snapshot.py:1310-1323).  Returns the file (or directory) name."""
        p = self.p
        params = dict(params or {})
        fmt = {'POS': 32, 'VEL': 32, 'ID': 'automatic'}
        fmt.update(dataformat or {})
        if snapformat not in (1, 2):
            raise ConceptGPUError(f'gadget_snapshot_params["snapformat"] = {snapformat} but must '
                                  'be 1 or 2')
        if not components:
            raise ConceptGPUError(f'Cannot save a {self.name} snapshot with no components')

        def get(c, key, default=None):
            return c.get(key, default) if isinstance(c, dict) else getattr(c, key, default)
        halo_like = [c for c in components if get(c, 'representation', 'particles') == 'particles'
                     and get(c, 'species') in ('matter', 'cold dark matter')]
        slots = [None]*num_particle_types
        for c in components:
            if get(c, 'representation', 'particles') != 'particles':
                continue  # only particle components are supported (snapshot.py:2478-2484)
            name = get(c, 'name')
            if name in component_names:
                slots[component_names.index(name)] = c
            elif len(halo_like) == 1 and c is halo_like[0]:
                slots[1] = c  # Mapping the component to "GADGET halo"
        comps = [c for c in slots if c is not None]
        if not comps:
            raise ConceptGPUError(f'No components left to store in the {self.name} snapshot')
        u = p.units
        H0 = params.get('H0', p.H0)
        a = params.get('a', a if a is not None else getattr(p, 'a_begin', 1.0))
        boxsize = params.get('boxsize', p.boxsize)
        Ωm = params.get('Ωm', p.Ωm)
        ΩΛ = params.get('ΩΛ', 1 - Ωm)
        h = H0/(100*u.km/(u.s*u.Mpc))
        self.h = h
        self.unit_length = _unit(self.unit_expr['length'], h, p)
        self.unit_velocity = _unit(self.unit_expr['velocity'], h, p)
        self.unit_mass = _unit(self.unit_expr['mass'], h, p)
        num = [0 if c is None else int(get(c, 'N')) for c in slots]
        hw = [n//2**32 for n in num]
        lw = [n - 2**32*w for n, w in zip(num, hw)]
        hd = {'Npart': list(num), 'Massarr': [0.0]*6, 'Time': a, 'Redshift': 1/a - 1, 'FlagSfr': 0,
              'FlagFeedback': 0, 'Nall': lw, 'FlagCooling': 0, 'NumFiles': 1,
              'BoxSize': boxsize/self.unit_length, 'Omega0': Ωm, 'OmegaLambda': ΩΛ,
              'HubbleParam': h, 'FlagAge': 0, 'FlagMetals': 0, 'NallHW': hw, 'flag_entr_ics': 0}
        for j, c in enumerate(slots):
            if c is not None:
                hd['Massarr'][j] = get(c, 'mass')/self.unit_mass
        for key, val in (header or {}).items():
            simple = key.lower().replace(' ', '').replace('-', '').replace('_', '')
            for k in hd:
                if k.lower().replace('_', '') == simple:
                    hd[k] = [type(x)(v) for v, x in zip(val, hd[k])] if isinstance(hd[k], list) \
                        else type(hd[k])(val)
                    break
        self.header = hd
        self.snapformat = snapformat
        ntot = sum(num)
        idbits = fmt['ID']
        if idbits == 'automatic':
            idbits = 64 if ntot > 2**32 else 32
        for k, b in (('POS', fmt['POS']), ('VEL', fmt['VEL']), ('ID', idbits)):
            if b not in (32, 64):
                raise ConceptGPUError(f'Could not understand gadget_snapshot_params["dataformat"]'
                                      f'[{k}] = {b}')

        def arrays(c):
            if isinstance(c, dict):
                return c['pos'], c['mom'], c.get('ids')
            ids = c.host('ids') if getattr(c, 'use_ids', False) else None
            return c.host('pos'), c.host('mom'), ids
        from . import comm
        active = comm.active()
        writer = active is None or active.rank == 0

        def block_bgn(f, size, name):
            if snapformat == 2:
                f.write(struct.pack('<I4sII', 8, name.ljust(4).encode('ascii'), 4 + size + 4, 8))
            f.write(struct.pack('<I', size))
        data = [arrays(c) for c in comps]   # (collective on several domains)
        # how many particles a file may hold: what a signed 32-bit block size leaves for the
        # wider of POS and VEL (snapshot.py:706-732), or the caller's number
        file_max = ((2**31 - 1) - 2*4)//(3*(max(fmt['POS'], fmt['VEL'])//8))
        if particles_per_file != 'automatic' and int(particles_per_file) > 0:
            file_max = int(particles_per_file)
        Ns = [int(get(c, 'N')) for c in comps]
        per_file = divvy(Ns, file_max)
        num_files = len(per_file)
        hd['NumFiles'] = num_files
        if 'numfiles' in {k.lower().replace(' ', '').replace('-', '').replace('_', '')
                          for k in (header or {})}:
            hd['NumFiles'] = int(next(v for k, v in header.items() if k.lower().replace(
                ' ', '').replace('-', '').replace('_', '') == 'numfiles'))
        if not writer:
            return filename
        types = [slots.index(c) for c in comps]
        id_bases = [sum(Ns[:i]) for i in range(len(comps))]
        if num_files > 1:
            if os.path.isfile(filename):
                os.remove(filename)
            os.makedirs(filename, exist_ok=True)
            # files of an earlier dump into this directory (snapshot.py:1003-1010)
            import glob as _glob
            import re as _re
            prefix = f'{filename}/{output_base}.'
            for old in _glob.glob(prefix + '*'):
                if _re.fullmatch(r'\d+', old[len(prefix):]):
                    os.remove(old)
        else:
            os.makedirs(os.path.dirname(os.path.abspath(filename)) or '.', exist_ok=True)
        done = [0]*len(comps)   # rows of each component already written to earlier files
        for file_index, counts in enumerate(per_file):
            fn = f'{filename}/{output_base}.{file_index}' if num_files > 1 else filename
            nfile = sum(counts)
            npart = [0]*num_particle_types
            for t, n in zip(types, counts):
                npart[t] = n
            with open(fn, 'wb') as f:
                block_bgn(f, headersize, 'HEAD')
                size = 0
                for key, fm in header_fields:
                    v = npart if key == 'Npart' else hd[key]
                    v = v if isinstance(v, list) else [v]
                    if fm.endswith('d'):
                        v = [correct_float(x) for x in v]
                    b = struct.pack('<' + fm, *v)
                    f.write(b)
                    size += len(b)
                f.write(b'\0'*(headersize - size))
                f.write(struct.pack('<I', headersize))
                for name, bits in (('POS', fmt['POS']), ('VEL', fmt['VEL'])):
                    size = nfile*3*(bits//8)
                    block_bgn(f, size, name)
                    for i, (c, (pos, mom, ids)) in enumerate(zip(comps, data)):
                        rows = slice(done[i], done[i] + counts[i])
                        if name == 'POS':
                            unit = self.unit_length
                            val = np.asarray(pos, dtype=np.float64).reshape(-1, 3)[rows]
                            val = val.reshape(-1)*(1/unit)
                            # safeguard against round-off: compared at the block's precision
                            # (snapshot.py:1117-1118, 1378-1391)
                            # (the value is cast FIRST: one that rounds up to the box size
                            # in single precision is wrapped to 0 there, as the reference's)
                            dt_ = np.float32 if bits == 32 else np.float64
                            box = dt_(boxsize/unit)
                            out = val.astype(dt_)
                            hi = out >= box
                            out[hi] -= box
                        else:
                            unit = self.unit_velocity*get(c, 'mass')*a**1.5
                            val = np.asarray(mom, dtype=np.float64).reshape(-1, 3)[rows]
                            val = val.reshape(-1)*(1/unit)
                            out = val.astype(np.float32 if bits == 32 else np.float64)
                        out.astype('<f4' if bits == 32 else '<f8').tofile(f)
                    f.write(struct.pack('<I', size))
                size = nfile*(idbits//8)
                block_bgn(f, size, 'ID')
                for i, (c, (pos, mom, ids)) in enumerate(zip(comps, data)):
                    if ids is None:  # running numbers, continued over the components
                        part = np.arange(id_bases[i] + done[i], id_bases[i] + done[i] + counts[i],
                                         dtype=np.uint64)
                    else:
                        part = np.asarray(ids)[done[i]:done[i] + counts[i]]
                    if len(part) and int(part.max()) >= 2**idbits:
                        # (snapshot.py:1063-1080 warns as well: the identifiers do not fit)
                        import warnings
                        warnings.warn(f'{get(c, "name")}: identifiers up to {int(part.max())} are '
                                      f'written as {idbits}-bit integers and wrap around; set '
                                      'gadget_snapshot_params["dataformat"]["ID"] = 64')
                    part.astype('<u4' if idbits == 32 else '<u8').tofile(f)
                f.write(struct.pack('<I', size))
            for i, n in enumerate(counts):
                done[i] += n
        return filename

    def to_components(self, device=None):
        """GPU Components (concept_amd.species.Component) holding the loaded particles"""
        from .species import Component
        out = []
        import torch
        for c in self.components:
            comp = Component(c['name'], c['species'], N=c['N'], mass=c['mass'], device=device)
            if comp.nprocs != self.nprocs:
                raise ConceptGPUError(
                    f'snapshot read as the share of rank {self.rank} of {self.nprocs}, but the '
                    f'active decomposition has {comp.nprocs} domains')
            if self.nprocs == 1:
                comp.populate(c['pos'], 'pos')
                comp.populate(c['mom'], 'mom')
                if c.get('ids') is not None:
                    comp.populate(c['ids'], 'ids')  # the file's ID block (snapshot.py:1573-1600)
            else:
                # this rank's rows of the file; exchange() (inside populate_local) re-homes
                # them.  Without an ID block the reference numbers the particles by their row in
                # the file (snapshot.py:2313-2322)
                dev = comp.device
                ids = c['ids'] if c.get('ids') is not None else \
                    np.arange(c['start_local'], c['start_local'] + c['N_local'], dtype=np.int64)
                comp.populate_local(torch.as_tensor(c['pos'], device=dev),
                                    torch.as_tensor(c['mom'], device=dev),
                                    torch.as_tensor(ids, device=dev),
                                    first_row=c['start_local'])
            out.append(comp)
        return out


def load(filename, only_params=False, params=None, units=None, rank=None, nprocs=None):
    """snapshot.load (snapshot.py:3120-3230) for GADGET files; over several domains every rank
    reads its own share (see the module docstring)"""
    return GadgetSnapshot(params, units, rank, nprocs).load(filename, only_params)


def save(components, filename, a=None, params=None, snapformat=2, dataformat=None, header=None,
         units=None, particles_per_file='automatic', output_base='snapshot'):
    """snapshot.save (snapshot.py:3060-3118) for snapshot_type = 'gadget'"""
    return GadgetSnapshot(None, units).save(components, filename, a, params, snapformat,
                                            dataformat, header, particles_per_file, output_base)
