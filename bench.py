#!/usr/bin/env python3
"""bench.py — PM steps/s and particle-updates/s of the MI355X gravity stepper.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one full PM base step of the reference's time loop
(main.py:335-358: drift -> long-range kick) on synthetic, HBM-resident
particles: drift (A11) + tile sort, CIC deposit (A1/A2), forward FFT (A4),
k-space Poisson kernel (A5/A6), inverse FFT (A8), fused finite-difference +
CIC gather + kick (A9/A10).  Workload at N=1: BASELINE.json's metric
configuration, 2^28 (~256M) particles on a 1024^3 mesh, FP64.  The particles
carry thermal momenta (rms displacement 0.2 mesh cells per step), so the tile
sort really reorders and, on N > 1 GPUs, particles really change domain.

--gpus N > 1: one process per GPU over RCCL.  Started either by an external
launcher (torch.distributed.run: WORLD_SIZE/RANK/LOCAL_RANK in the
environment) or, with WORLD_SIZE unset, by this script itself, which then
re-executes under torch.distributed.run on 127.0.0.1.  With fewer visible GPUs
than ranks the ranks share the GPUs and exchange through host memory (gloo): a
functional run of the sharded path, flagged as such in the JSON line.

Prints ONE JSON line (rank 0), the last line of stdout.  `roofline` is for the
dominant hand-written kernel, from HIP events on the stream the kernels run
on; `cpu_baseline` is the C oracle (oracle/, a port of the reference's
algorithm) timed on a bounded sample on this host's cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
XGMI_LINK_GBS_DIR = 76.8     # one xGMI link, one direction (153.6 GB/s bidirectional; 7 links)
def _latest(pattern):
    """the newest round's committed recording of that name (profiles/rNN_...)"""
    import glob
    found = sorted(glob.glob(os.path.join(REPO, 'profiles', pattern)))
    return os.path.relpath(found[-1], REPO) if found else os.path.join('profiles', pattern)


PMC_FILE = _latest('r??_pmc_hbm_traffic.json')

WORKLOADS = {
    # name: (particles, gridsize)
    'ns_256M_1024': (2**28, 1024),       # BASELINE.json metric config
    'c2_256c_512': (256**3, 512),        # configs[1]
    'c1_128c_256': (128**3, 256),        # configs[0]
    'c3_1024c_2048': (1024**3, 2048),    # configs[3] (meant for 8 GPUs; fits one: ~175 GB)
    'c4_512c_1024': (512**3, 1024),      # configs[4]'s particle part
    'tiny': (32**3, 64),
}


# --weak: the work PER GPU stays the same as the ranks grow — 2^25 particles and ~1.3e8 mesh
# cells each (a cubic mesh cannot be scaled by exactly 2 and 4: 640^3 and 800^3 are within 5 %
# of twice and four times 512^3; they are not powers of two, so their slab transforms take the
# rocFFT backend, DESIGN.md §6 — stated in the line as fft_backend).  P = 8 is the metric's
# own size.
WEAK = {1: (2**25, 512), 2: (2**26, 640), 4: (2**27, 800), 8: (2**28, 1024)}


def survey_bytes(n_p, n_g):
    """algorithmic bytes of the reference's UNFUSED phases (what its table
    credits a phase with).  Reported beside the real figures, never turned into a fraction of
    peak for a single fused kernel."""
    return {
        'deposit': 24*n_p + 8*n_g,
        'poisson': 48*n_g + 16*n_g + 48*n_g,          # A4 + A5-A7 + A8
        'fft_zy_forward_chunked': 32*n_g, 'fft_yz_backward_chunked': 32*n_g,
        'fft_z_forward': 16*n_g, 'fft_y_forward': 16*n_g, 'fft_y_backward': 16*n_g,
        'fft_z_backward': 16*n_g,
        'fft_x_fused_kspace': 48*n_g,                 # x forward + k-space + x inverse
        'gather_kick': 72*n_p + 24*n_g,               # row A10: three force grids
        'drift': 72*n_p,
        'drift_sort': 72*n_p + 120*n_p,               # drift row + a sort that re-reads pos
        'kick_drift_sort': 72*n_p + 24*n_g + 72*n_p + 120*n_p,  # rows A10 + A11 + the sort
        'sr_cells': 56*n_p, 'sr_sweep': 72*n_p,
    }


def moved_bytes(n_p, n_g, with_ids=False):
    """Bytes each kernel of THIS build must move between L2 and the memory fabric as designed
     — the quantity rocprofv3's FETCH_SIZE + WRITE_SIZE measure and the one
    every `frac` below is computed from."""
    ids = 16*n_p if with_ids else 0
    return {
        'zero': 8*n_g,
        'deposit': 24*n_p + 8*n_g,                    # read pos, write the mesh once
        'poisson': 80*n_g,                            # five in-place passes
        'fft_forward': 48*n_g, 'kspace': 16*n_g, 'fft_backward': 48*n_g,
        'fft_zy_forward_chunked': 32*n_g, 'fft_yz_backward_chunked': 32*n_g,
        'fft_z_forward': 16*n_g, 'fft_y_forward': 16*n_g, 'fft_y_backward': 16*n_g,
        'fft_z_backward': 16*n_g,
        'fft_x_fused_kspace': 16*n_g,                 # one read + one write of the mesh
        'gather_kick': 72*n_p + 8*n_g,                # pos, mom RMW, the potential once
        'drift': 72*n_p,
        'kick_drift_sort': 96*n_p + 8*n_g + ids,      # pos+mom read, pos+mom written at their
                                                      # new places, the potential once
        'drift_sort': 96*n_p + ids,                   # pos+mom read, pos+mom written (prepared
                                                      # histogram: no separate counting pass)
        'sort': 96*n_p + 24*n_p + ids,
        'sr_cells': 2*24*n_p + 8*n_p,
        'sr_sweep': 24*n_p + 48*n_p,                  # not HBM-bound: FP64 pair arithmetic
    }


# ---------------------------------------------------------------------------
# CPU baseline (the oracle; test infrastructure used here only as the thing timed beside)
# ---------------------------------------------------------------------------
def cpu_baseline(workload):
    """The oracle (a C port of the reference's algorithm) timed on this host's cores, in a
    process of its own: the OpenMP runtime reads its binding from the environment when it
    starts (threads spread over the NUMA domains and pinned), and the ~50 GB of the
    north-star-size sample go back to the system before the line is printed."""
    env = dict(os.environ)
    env.setdefault('OMP_PROC_BIND', 'spread')
    env.setdefault('OMP_PLACES', 'threads')
    env['OMP_DYNAMIC'] = 'false'
    # the host threads this process may use, counted HERE: in the child the OpenMP runtime binds
    # the initial thread to its first place when the library is loaded (OMP_PROC_BIND), after
    # which sched_getaffinity() there answers 1
    try:
        env['CONCEPT_BENCH_HOST_THREADS'] = str(len(os.sched_getaffinity(0)))
    except AttributeError:
        env['CONCEPT_BENCH_HOST_THREADS'] = str(os.cpu_count() or 1)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-child',
                            workload], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=1500)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
        if r.returncode or not lines:
            return {'value': None, 'error': r.stderr.decode()[-400:]}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired:
        return {'value': None, 'error': 'cpu baseline timed out'}


def cpu_baseline_child(workload):
    """Runs in its own process (see cpu_baseline).  Same workload as the GPU line when the host
    has the memory for it (2^28 particles / 1024^3 mesh: ~50 GB): the thread count is chosen AT
    THAT SIZE — candidates ranked on the 256^3 / 512^3 sample first, the best two and "all
    threads" then each take one full PM step of the north-star size; the fastest is the
    figure.  OpenMP over the particle and plane loops (slab-privatised deposit: no atomics,
    like the reference's one-domain-per-core ranks); FFT: scipy's pocketfft with as many
    workers (the reference's FFTW-MPI is not available here).  Single-thread and 256^3 / 512^3
    figures are kept as extras."""
    import numpy as np
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = None
    from oracle import oracle
    oracle.build()
    n_p, N = WORKLOADS[workload]

    import ctypes
    omp_lib = oracle.lib('omp')
    # Loading the OpenMP runtime has bound THIS thread to its first place (OMP_PROC_BIND): the
    # worker threads scipy.fft starts later inherit the caller's mask and would all share that one
    # hardware thread (the transforms then take as long with 32 workers as with one).  The
    # OpenMP workers keep their places; the initial thread gets the process's mask back.
    if allowed:
        os.sched_setaffinity(0, allowed)
    _dp = ctypes.POINTER(ctypes.c_double)
    omp_lib.orc_fill_tiled.argtypes = [_dp, ctypes.c_int64, _dp, ctypes.c_int64, ctypes.c_double,
                                       ctypes.c_double]
    omp_lib.orc_fill_tiled.restype = None

    def particles(n, L):
        """positions uniform in the box, momenta 0.2 cells rms per step like the GPU run.  Both
        arrays are WRITTEN BY THE OPENMP THREADS (static schedule, as every particle loop of
        the port): first touch spreads their pages over the NUMA domains — generated by one
        numpy thread they would all sit on one, and no thread count beyond that domain's
        cores would help.  The values are a block of 2^22 random rows repeated with a shift
        (positions) / as they are (momenta): they do not matter to the timing."""
        rng = np.random.default_rng(7)
        nb = min(n, 1 << 22)
        bpos = np.ascontiguousarray(rng.random((nb, 3))*L)
        bmom = np.ascontiguousarray(rng.normal(0, 0.2/3**0.5*1e3, (nb, 3)))
        pos = np.empty((n, 3), dtype=np.float64)
        mom = np.empty((n, 3), dtype=np.float64)
        ptr = lambda a: a.ctypes.data_as(_dp)
        omp_lib.orc_fill_tiled(ptr(pos), 3*n, ptr(bpos), 3*nb, 0.6180339887498949*L, L)
        omp_lib.orc_fill_tiled(ptr(mom), 3*n, ptr(bmom), 3*nb, 0.0, 0.0)
        return pos, mom

    def run(fast, pos, mom, grid, nsteps, timings=None):
        L = float(grid)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            t1 = time.perf_counter()
            oracle.drift(pos, mom, 1e-3, L, fast=fast)
            if timings is not None:
                timings['drift'] = timings.get('drift', 0.0) + time.perf_counter() - t1
            oracle.pm_long_range(pos, mom, mass=1.0, boxsize=L, gridsize=grid,
                                 G_Newton=1.0, dt_1=1e-3, dt_dens=1e-3, dt_kick=1e-3,
                                 diff_order=2, fast=fast, want_indices=False, timings=timings)
        return time.perf_counter() - t0

    def phase_table(timings, nsteps, n, grid):
        """the CPU path's phases beside the GPU's — wall time per step and the
        rate at which each moves SURVEY.md §8(d)'s algorithmic bytes"""
        sv = survey_bytes(n, grid**3)
        credit = {'deposit': sv['deposit'], 'fft_forward': 48*grid**3, 'kspace': 16*grid**3,
                  'fft_backward': 48*grid**3, 'gather_kick': sv['gather_kick'],
                  'drift': sv['drift']}
        out = {}
        for k, v in timings.items():
            e = {'s_per_step': round(v/nsteps, 4)}
            if k in credit:
                e['GBps_of_survey_8d_bytes'] = round(credit[k]/(v/nsteps)/1e9, 2)
            out[k] = e
        return out

    avail = int(os.environ.get('CONCEPT_BENCH_HOST_THREADS', '0'))
    if avail < 1:  # (called by hand: the count may already be the one thread OpenMP left)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
    mem_gb = 0.0
    try:
        with open('/proc/meminfo') as f:
            for line in f:
                if line.startswith('MemAvailable'):
                    mem_gb = int(line.split()[1])/2**20
    except OSError:
        pass
    omp = omp_lib
    flags = '-O3 -funroll-loops -ffast-math (reference src/Makefile flags)'
    binding = f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}"
    # 1. candidates ranked at 256^3 / 512^3 (2 steps each after one warm-up step)
    pos2, mom2 = particles(256**3, 512.0)
    trial = {}
    for threads in sorted({min(avail, t) for t in (16, 32, 64, 128, avail//2, avail)} - {0}):
        omp.orc_threads(threads)
        run('omp', pos2, mom2, 512, 1)
        trial[threads] = run('omp', pos2, mom2, 512, 2)/2
    ranked = sorted(trial, key=trial.get)
    c2_threads = ranked[0]
    omp.orc_threads(c2_threads)
    tm2 = {}
    run('omp', pos2, mom2, 512, 2, tm2)
    c2 = {'value': 256**3/trial[c2_threads], 'unit': 'particle-updates/s', 'cores': c2_threads,
          'steps_per_sec': 1/trial[c2_threads], 'phases': phase_table(tm2, 2, 256**3, 512),
          'sample': '256^3 particles / 512^3 mesh (BASELINE configs[1] size), 2 PM steps'}
    del pos2, mom2
    # 2. one core, numpy's pocketfft (the reference's own pure-Python FFT)
    pos1, mom1 = particles(128**3, 256.0)
    steps_one = 4
    dt_one = run(True, pos1, mom1, 256, steps_one)
    del pos1, mom1
    single = {'value': 128**3*steps_one/dt_one, 'unit': 'particle-updates/s', 'cores': 1,
              'steps_per_sec': steps_one/dt_one,
              'sample': f'128^3 particles / 256^3 mesh (BASELINE configs[0]), {steps_one} PM '
                        f'steps, oracle C port built {flags} + numpy pocketfft, '
                        f'{dt_one:.1f} s wall'}
    # what this host's memory gives those thread counts: STREAM triad (a = b + s c, 3 x 2^27
    # doubles touched first by the threads that sweep them, best of 3) — the ceiling beside
    # which the port's phases and its scaling with threads are to be read
    omp.orc_stream_triad.restype = ctypes.c_double
    omp.orc_stream_triad.argtypes = [_dp, _dp, _dp, ctypes.c_int64, ctypes.c_int]
    triad = {}
    nt_ = 1 << 27
    for threads in sorted({min(avail, t) for t in (1, 32, 128, avail)} - {0}):
        omp.orc_threads(threads)
        arrs = [np.empty(nt_, dtype=np.float64) for _ in range(3)]
        dt_ = omp.orc_stream_triad(*(a_.ctypes.data_as(_dp) for a_ in arrs), nt_, 3)
        triad[str(threads)] = round(24*nt_/dt_/1e9, 1)
        del arrs
    extras = {'host_stream_triad_GBps_by_threads': triad,
              'thread_scaling_note': (
                  'the fastest trial is the one reported; where more threads are slower the port '
                  'is held by the host\'s memory system, not by its loops: compare the phases\' '
                  'GB/s (SURVEY.md §8(d) bytes) with the triad figures — a thread count beyond '
                  'the one that saturates the triad adds contention (scattered 8-byte updates of '
                  'the deposit and gather across NUMA domains), no bandwidth'),
              'thread_trials_s_per_step_256c_512': {str(k): round(v, 3) for k, v in trial.items()},
              'c2_256c_512': c2, 'single_thread': single, 'host_threads': avail,
              'host_mem_available_GB': round(mem_gb, 1), 'binding': binding,
              'fft_backend': 'scipy.fft (pocketfft), workers = OpenMP threads'}
    need_gb = (48*n_p + 6*8*(N + 4)**3)/2**30 + 8
    if (n_p, N) == (256**3, 512) or mem_gb < need_gb:
        # the workload IS the 256^3 sample, or this host cannot hold the workload
        out = dict(c2, kind='port', **extras)
        out['sample'] = (f'{c2["sample"]} = {("the workload " + workload) if (n_p, N) == (256**3, 512) else "NOT the workload: host has " + format(mem_gb, ".0f") + " GB available, the " + workload + " sample needs " + format(need_gb, ".0f")}'
                         f'; oracle C port built {flags} -fopenmp on {c2_threads} of {avail} '
                         f'host threads ({binding}) + scipy.fft with as many workers')
        print(json.dumps(out))
        return
    # 3. the workload itself: one PM step per candidate thread count — bounded: what a step at
    # size will take is known from the sample (the port is memory-bound: time ~ particles + cells)
    est = trial[c2_threads]*(n_p/256**3)
    if est > 60:
        # (a host that offers this process a handful of threads: a full step would take minutes)
        out = dict(c2, kind='port', **extras)
        out['sample'] = (f'{c2["sample"]} = 1/{n_p//256**3} of the workload {workload} (same code, same work '
                         f'per particle and cell): a full step at size would take ~{est:.0f} s on the '
                         f'{avail} host thread(s) this process may use, beyond the bound of the '
                         f'baseline leg; oracle C port built {flags} -fopenmp on {c2_threads} of '
                         f'{avail} host threads ({binding}) + scipy.fft with as many workers')
        print(json.dumps(out))
        return
    pos, mom = particles(n_p, float(N))
    cand = []
    for t in ([ranked[0]] if est > 40 else [ranked[0], min(avail, 128), avail]):
        if t not in cand:
            cand.append(t)
    at_size, at_size_phases = {}, {}
    for threads in cand:
        omp.orc_threads(threads)
        tm = {}
        at_size[threads] = run('omp', pos, mom, N, 1, tm)
        at_size_phases[threads] = phase_table(tm, 1, n_p, N)
    cores = min(at_size, key=at_size.get)
    dt = at_size[cores]
    out = {'value': n_p/dt, 'unit': 'particle-updates/s', 'cores': cores, 'kind': 'port',
           'steps_per_sec': 1/dt,
           'sample': f'{workload}: {n_p} particles / {N}^3 mesh — the GPU line\'s workload — one '
                     f'full PM step (drift, CIC deposit, FFT Poisson solve, FD gradient + CIC '
                     f'gather-kick) per candidate thread count, fastest reported: oracle C port '
                     f'built {flags} -fopenmp (slab-privatised deposit, no atomics) on {cores} '
                     f'of {avail} host threads ({binding}) + scipy.fft with {cores} workers, '
                     f'{dt:.1f} s wall per step',
           'thread_trials_s_per_step_at_size': {str(k): round(v, 2) for k, v in at_size.items()},
           'phases': at_size_phases[cores],
           'phases_note': ('wall time per step of each phase of the port at the reported thread '
                           'count; GBps = SURVEY.md §8(d) bytes of the phase / its time (zero, '
                           'slab_decompose and domain_decompose are layout passes the GPU build '
                           'does not have)')}
    out.update(extras)
    print(json.dumps(out))


# ---------------------------------------------------------------------------
# synthetic particles
# ---------------------------------------------------------------------------
def make_positions(torch, args, n_p, N, L, dev, gen, mesh=None):
    pos = torch.rand((n_p, 3), dtype=torch.float64, device=dev, generator=gen)
    if args.dist == 'uniform':
        pos.mul_(L)
    elif args.dist == 'lattice':
        # a cubic lattice; a count that is not a cube fills the first n_p sites of the next
        # larger one (its last layers stay partly empty)
        side = round(n_p**(1/3))
        if side**3 < n_p:
            side += 1
        idx = torch.arange(n_p, device=dev)
        lat = torch.stack([idx//(side*side), (idx//side) % side, idx % side], 1).double()
        disp = torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*1.5*(L/N)
        pos = torch.remainder((lat + 0.5)*(L/side) + disp, L)
        del idx, lat, disp
    elif args.dist == 'zeldovich':
        # a cubic lattice displaced by a Gaussian random field with a k^-2
        # spectrum, 3-D rms displacement 1.5 mesh cells — made with the build's own deposit, FFT
        # and gather: the shot noise of n_p random points is white, its potential (one Poisson
        # solve) has a k^-4 spectrum and the potential's gradient at the lattice sites, k^-2
        if mesh is None:
            raise SystemExit('--dist zeldovich: single-GPU runs only')
        side = round(n_p**(1/3))
        if side**3 < n_p:
            side += 1
        pos.mul_(L).clamp_(max=float(torch.nextafter(torch.tensor(L, dtype=torch.float64),
                                                     torch.tensor(0.0, dtype=torch.float64))))
        mesh.zero()
        mesh.deposit(pos, 1.0)
        mesh.poisson_solve(4, 1.0, False, 0.0)
        idx = torch.arange(n_p, device=dev)
        pos = torch.stack([idx//(side*side), (idx//side) % side, idx % side], 1).double()
        pos.add_(0.5).mul_(L/side)
        del idx
        disp = torch.zeros_like(pos)
        mesh.gather_kick(pos, disp, 2, 1.0)
        disp.mul_(1.5*(L/N)/float(disp.square().sum(1).mean().sqrt()))
        pos = torch.remainder(pos.add_(disp), L)
        del disp
    else:  # clustered: 64 Gaussian blobs of sigma = L/40 holding 80 % of the particles
        centres = torch.rand((64, 3), dtype=torch.float64, device=dev, generator=gen)*L
        which = torch.randint(0, 64, (n_p,), device=dev, generator=gen)
        blob = centres[which] + torch.randn((n_p, 3), dtype=torch.float64, device=dev,
                                            generator=gen)*(L/40)
        keep = torch.rand(n_p, dtype=torch.float64, device=dev, generator=gen) < 0.2
        pos = torch.where(keep[:, None], pos*L, torch.remainder(blob, L))
        del centres, which, blob, keep
    top = float(torch.nextafter(torch.tensor(L, dtype=torch.float64),
                                torch.tensor(0.0, dtype=torch.float64)))
    pos.clamp_(min=0.0, max=top)
    return pos


def thermal_momenta(torch, args, shape, cell, mass, dt, dev, gen):
    """Maxwellian momenta with a 3-D rms displacement of --thermal mesh cells per step."""
    if args.thermal <= 0:
        return torch.zeros(shape, dtype=torch.float64, device=dev)
    sigma = args.thermal/3**0.5*cell*mass/dt
    mom = torch.randn(shape, dtype=torch.float64, device=dev, generator=gen)
    return mom.mul_(sigma)


def pmc_traffic(dom_kernel, workload):
    """HBM (L2 <-> fabric) bytes per launch of the dominant kernel from the committed PMC run of
    this command (rocprofv3 --pmc needs its own process: it cannot be collected from inside the
    bench).  (None, why) unless the file describes this workload, this kernel AND the sources
    the loaded library was built from (tools/pmc_traffic.py stamps their hash; the build writes
    it beside the .so): a counter run of other kernels is not quoted."""
    try:
        pmc = json.load(open(os.path.join(REPO, PMC_FILE)))
    except Exception:
        return None, f'{PMC_FILE} not found'
    if pmc.get('workload_name') != workload:
        return None, f'{PMC_FILE} describes workload {pmc.get("workload_name")}, not {workload}'
    try:
        from concept_amd import build as cg_build
        built = open(cg_build.LIB + '.srchash').read().strip()
    except Exception:
        built = None
    if built is None or pmc.get('csrc_hash') != built:
        return None, (f'{PMC_FILE} was collected on a library built from sources '
                      f'{pmc.get("csrc_hash")}, the one loaded now is built from {built}: not '
                      'quoted (re-collect with tools/record_profiles.sh)')
    for kname, entry in pmc.get('kernels', {}).items():
        if entry.get('bench_key') == dom_kernel:
            return entry['total_GB']*1e9, (
                f"{PMC_FILE}: kernel {kname}, collected at commit {pmc.get('commit', '?')} "
                f"({pmc.get('date', '?')}, sources {built}) with `{pmc.get('command', '?')}`; "
                'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 corrections '
                'of MI355X_MICROARCH.md')
    return None, f'{PMC_FILE} has no entry for {dom_kernel}'


# ---------------------------------------------------------------------------
# particles as a function of (seed, global identifier) only — the same box on 1, 2, 4, 8 ranks —
# and the check an N-rank run carries with it
# ---------------------------------------------------------------------------
ID_CHUNK = 1 << 20
VERIFY_SAMPLES = 512
VERIFY_DIR = os.path.join(REPO, '.bench_verify')
VERIFY_COMMITTED = 'profiles/r??_bench_verify_{key}.json'   # (the newest round's)


def global_particles(torch, args, n_p, L, cell, mass, dt, dev, rank=0, world=1, mesh=None):
    """(pos, mom, ids) of this rank's share of the box: uniform positions, Maxwellian momenta
    (--thermal), each chunk of ID_CHUNK identifiers from a generator seeded by (seed, chunk) —
    the same particles whatever the number of ranks.  With a slab `mesh` (N > 1 ranks) every
    rank walks ALL chunks and keeps the particles its slab owns (cg_owner_rank, the exchange's own
    rule): nobody has to be shipped at set-up; without one, the chunks rank, rank + world, ..."""
    sigma = args.thermal/3**0.5*cell*mass/dt if args.thermal > 0 else 0.0
    top = float(torch.nextafter(torch.tensor(L, dtype=torch.float64),
                                torch.tensor(0.0, dtype=torch.float64)))
    nchunks = (n_p + ID_CHUNK - 1)//ID_CHUNK
    by_owner = mesh is not None and world > 1
    mine = range(nchunks) if by_owner else range(rank, nchunks, world)
    if not by_owner:
        # the count is known: write the chunks in place (2^30 particles leave no room for a copy)
        m_total = sum(min(ID_CHUNK, n_p - c*ID_CHUNK) for c in mine)
        pos = torch.empty((m_total, 3), dtype=torch.float64, device=dev)
        mom = torch.zeros((m_total, 3), dtype=torch.float64, device=dev)
        ids = torch.empty(m_total, dtype=torch.int64, device=dev)
        at = 0
        for c in mine:
            m = min(ID_CHUNK, n_p - c*ID_CHUNK)
            gen = torch.Generator(device=dev).manual_seed(1000003*args.seed + c)
            x = torch.rand((ID_CHUNK, 3), dtype=torch.float64, device=dev, generator=gen)
            pos[at:at + m] = x[:m].mul_(L).clamp_(min=0.0, max=top)
            if sigma:
                v = torch.randn((ID_CHUNK, 3), dtype=torch.float64, device=dev, generator=gen)
                mom[at:at + m] = v[:m].mul_(sigma)
            ids[at:at + m] = torch.arange(c*ID_CHUNK, c*ID_CHUNK + m, device=dev)
            at += m
        return pos, mom, ids
    P, M, I = [], [], []
    for c in mine:
        m = min(ID_CHUNK, n_p - c*ID_CHUNK)
        gen = torch.Generator(device=dev).manual_seed(1000003*args.seed + c)
        # (always a whole chunk: how a generator's stream maps onto the elements may depend on
        # the tensor's size)
        x = torch.rand((ID_CHUNK, 3), dtype=torch.float64, device=dev, generator=gen)
        x = x[:m].mul_(L).clamp_(min=0.0, max=top)
        v = None
        if sigma:
            v = torch.randn((ID_CHUNK, 3), dtype=torch.float64, device=dev, generator=gen)
            v = v[:m].mul_(sigma)
        ids = torch.arange(c*ID_CHUNK, c*ID_CHUNK + m, device=dev)
        if by_owner:
            keep = mesh.owner_rank(x.contiguous()) == rank
            x, ids = x[keep], ids[keep]
            v = v[keep] if v is not None else None
        P.append(x)
        M.append(v if v is not None else torch.zeros_like(x))
        I.append(ids)
    if not P:
        z = torch.zeros((0, 3), dtype=torch.float64, device=dev)
        return z, z.clone(), torch.zeros(0, dtype=torch.int64, device=dev)
    return torch.cat(P).contiguous(), torch.cat(M).contiguous(), torch.cat(I).contiguous()


def verify_key(args, name, steps_total):
    return f'{name}_seed{args.seed}_thermal{args.thermal:g}_steps{steps_total}'


def verify_replay(torch, args, domain, name, n_p, N, L, dev, rank, world, steps_total):
    """The sharded step sequence on the (seed, identifier)-defined box, with the identifiers
    travelling: tile sort, then steps_total x (deposit, solve, kick + drift + sort) — the
    sequence of the timed region (warm-up included) — and what it leaves behind: particle
    count, sum of mom^2, and the positions and momenta of the VERIFY_SAMPLES particles whose
    identifiers are multiples of n_p / VERIFY_SAMPLES.  Untimed."""
    from concept_amd.distributed import ParticleStore, RegionParticles, pm_step_regions
    mass, G, dt = 1.0, 1.0, 1e-4
    pos, mom, ids = global_particles(torch, args, n_p, L, L/N, mass, dt, dev, rank, world,
                                     mesh=domain.mesh)
    parts = ParticleStore(domain, pos, mom, ids, slack=1.15)
    del pos, mom, ids
    parts.exchange()
    parts.tile_sort()
    rp = RegionParticles(parts, slack=1.15)
    del parts
    contribution = mass*(float(N)**(-3)*(N/L)**3)
    C = -L**2*G/3.141592653589793
    for _ in range(steps_total):
        pm_step_regions(domain, rp, contribution, 4, C, mass*(-dt), dt/mass, diff_order=2)
    rp.check()
    cols = rp.columns()
    stride = max(n_p//VERIFY_SAMPLES, 1)
    sel = (cols['ids'] % stride == 0) & (cols['ids'] < stride*VERIFY_SAMPLES)
    out = {'n_local': int(cols['ids'].numel()),
           'sum_mom2': float(cols['mom'].square().sum()),
           'ids': cols['ids'][sel].cpu().numpy(), 'pos': cols['pos'][sel].cpu().numpy(),
           'mom': cols['mom'][sel].cpu().numpy()}
    del cols, rp
    torch.cuda.empty_cache()
    return out


def verify_save(key, ids, pos, mom, n, sum_mom2, extra):
    """the 1-rank values, where the N-rank runs of this box look for them"""
    os.makedirs(VERIFY_DIR, exist_ok=True)
    path = os.path.join(VERIFY_DIR, key + '.json')
    with open(path, 'w') as f:
        json.dump(dict(extra, key=key, particles=int(n), sum_mom2=float(sum_mom2).hex(),
                       ids=[int(i) for i in ids],
                       pos=[[float(v).hex() for v in row] for row in pos],
                       mom=[[float(v).hex() for v in row] for row in mom]), f)
    return path


def verify_load(key):
    import numpy as np
    for path in (os.path.join(VERIFY_DIR, key + '.json'),
                 os.path.join(REPO, _latest(os.path.basename(VERIFY_COMMITTED.format(key=key))))):
        if os.path.exists(path):
            d = json.load(open(path))
            unhex = lambda rows: np.array([[float.fromhex(v) for v in row] for row in rows])
            return {'path': os.path.relpath(path, REPO), 'particles': d['particles'],
                    'sum_mom2': float.fromhex(d['sum_mom2']), 'ids': np.array(d['ids']),
                    'pos': unhex(d['pos']), 'mom': unhex(d['mom']),
                    'made_by': d.get('made_by')}
    return None


# ---------------------------------------------------------------------------
# N > 1: x-slab domains
# ---------------------------------------------------------------------------
# one GPU, 2^28 particles / 1024^3 (profiles/r05_bench_ns_full_default.json), ms
LINK_MODEL_SINGLE = {'deposit': 3.1, 'fft_zy_pairs': 11.6, 'fft_x_fused': 4.0,
                     'kick_drift_sort': 9.0}


def link_model_predict(N, total, world, transform_bytes=None):
    """The link model of DESIGN.md §6: local kernels at 1/P of their single-GPU times, each
    transpose at (local transform)/P bytes per peer over that peer's own xGMI link, the z / y
    transforms hidden under the links when those are slower (the transposes are pipelined with
    them), the x pass not.  transform_bytes: a rank's transform buffer (default: its slab's
    N/P x N x (N + 2) doubles)."""
    if transform_bytes is None:
        transform_bytes = N*N*(N + 2)*8//world
    scale = (N/1024.0)**3/world
    local = {k: v*scale*((total/2.0**28)/(N/1024.0)**3 if k in ('deposit', 'kick_drift_sort')
                         else 1.0) for k, v in LINK_MODEL_SINGLE.items()}
    per_peer = transform_bytes/world
    t_transpose = per_peer/(XGMI_LINK_GBS_DIR*1e9)*1e3 if world > 1 else 0.0
    poisson_pred = max(2*t_transpose, local['fft_zy_pairs']) + local['fft_x_fused']
    pred = local['deposit'] + poisson_pred + local['kick_drift_sort']
    return {
        'single_gpu_kernel_ms': dict(LINK_MODEL_SINGLE),
        'local_kernel_ms_at_this_P': {k: round(v, 3) for k, v in local.items()},
        'bytes_per_peer_per_transpose': int(per_peer),
        'transpose_ms_at_link_rate': round(t_transpose, 3),
        'predicted_poisson_stage_ms': round(poisson_pred, 3),
        'predicted_step_ms': round(pred, 3),
        'predicted_particle_updates_per_s': round(total/(pred*1e-3), 1),
        'note': ('prediction = deposit/P + max(2 transposes at one xGMI link per peer '
                 f'({XGMI_LINK_GBS_DIR} GB/s one way), z/y transforms/P) + x pass/P + fused '
                 'particle pass/P; exchange of leavers and halo layers not priced (MBs)')}


def main_distributed(args, name, n_p, N, L, dev, rank, world, backend):
    """Strong scaling: the same total workload on `world` x-slab domains, one per GPU
    (concept_amd/distributed.py).  A step = drift + particle exchange + tile sort (fused:
    DistributedParticles.drift_exchange_sort) + long-range kick (deposit, ghost fold, FFT
    with two all-to-all transposes, ghost fill, gather-kick)."""
    import torch
    import torch.distributed as dist
    from concept_amd.distributed import (DistributedParticles, RegionParticles, SlabDomain,
                                         pm_kick, pm_step_regions)
    dom = SlabDomain(N, L, device=dev)
    # the box is a function of (seed, global identifier): the same particles on any number of
    # ranks (every rank walks all chunks of identifiers and keeps what its slab owns)
    cell = L/N
    mass, G, dt = 1.0, 1.0, 1e-4
    pos, mom, _ = global_particles(torch, args, n_p, L, cell, mass, dt, dev, rank, world,
                                   mesh=dom.mesh)
    parts = DistributedParticles(dom, pos, mom, None, slack=1.15)
    del pos, mom
    parts.exchange()
    parts.tile_sort()
    contribution = mass*(float(N)**(-3)*(N/L)**3)
    C = -L**2*G/3.141592653589793
    fused = not args.no_fused
    if fused:
        # kick + drift + tile sort in one pass over particles kept in tile regions with gaps;
        # the pass hands the leavers of the slab over, the next step ships them before its
        # deposit (the same cycle as the unfused step, entered after the sort)
        rp = RegionParticles(parts, slack=1.15)
        del parts
        parts = rp

    stage_events = []  # per timed step: [(name, event), ...] on the compute stream

    def step(record=False):
        evs = []

        def mark(name):
            if record:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append((name, e))
        mark('start')
        if fused:
            pm_step_regions(dom, parts, contribution, 4, C, mass*(-dt), dt/mass, diff_order=2,
                            mark=mark)
        else:
            # emigrants of the coming drift are shipped first, then one drift + sort pass pair;
            # the gather-kick histograms the tiles of the next drift
            parts.drift_exchange_sort(dt/mass)
            mark('drift+exchange+sort')
            pm_kick(dom, parts, contribution, 4, C, mass*(-dt), diff_order=2,
                    next_dt_over_mass=dt/mass, mark=mark)
        if record:
            stage_events.append(evs)

    for _ in range(args.warmup):
        step()
    parts.check()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    emig0 = parts.emigrants_total
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    parts.check()
    # this rank's stage times (the exchanges wait for the slowest peer inside their stage)
    stages = {}
    for evs in stage_events:
        for (_, a), (sname, b) in zip(evs[:-1], evs[1:]):
            stages[sname] = stages.get(sname, 0.0) + a.elapsed_time(b)/len(stage_events)
    dry = None
    if args.dry_links and world > 1:
        # The schedule with sleeping links against its two parts, all three measured the same
        # way (every rank at once, behind a barrier): the solve with the sleeps under its
        # transforms, the sleeps alone (what was requested), the transforms alone (the same
        # solve at an infinite rate).
        def solves(rate):
            dom.comm.dry_rate = rate
            dom.comm.dry_ms = 0.0
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                dom.mesh.poisson_solve(4, C, False, 0.0, fill=True)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1)/3, dom.comm.dry_ms/3
        solves(args.dry_links*1e9)
        with_ms, links_ms = solves(args.dry_links*1e9)
        compute_ms, _ = solves(1e30)
        hidden = compute_ms + links_ms - with_ms
        dry = {'rate_GBps_per_link': args.dry_links, 'links_ms_per_solve': round(links_ms, 3),
               'transforms_alone_ms': round(compute_ms, 3), 'solve_ms': round(with_ms, 3),
               'hidden_ms': round(hidden, 3),
               'overlap_fraction': round(hidden/max(min(compute_ms, links_ms), 1e-9), 3),
               'note': ('every FFT transpose replaced by a device sleep of (bytes to one peer) / '
                        'rate on a side stream; overlap = (transforms alone + sleeps - solve) / '
                        'min(transforms alone, sleeps).  Ranks sharing one GPU slow each '
                        'other\'s transforms down: the fraction describes the schedule, the '
                        'milliseconds do not describe a node')}
    red = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(red, op=dist.ReduceOp.MAX)
    elapsed = float(red.item())
    # what the timed run left behind, rank by rank: particles (load balance of the slabs) and
    # the sum of mom^2
    m2 = parts.measure_momentum() if fused else dom.mesh.measure_momentum(parts.view('mom'))
    mine = torch.tensor([float(parts.n), float(m2[0]), float(rank),
                         float(torch.cuda.current_device())], dtype=torch.float64, device=dev)
    seen = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    seen = torch.stack(seen).cpu().numpy()
    emig_total = parts.emigrants_total
    # outside the timed region: what one whole FFT transpose costs on this transport by itself
    # (one all_to_all_single of the transpose buffer), for reading the stage times above
    probe = None
    if world > 1:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dom.comm.all_to_all(dom.tbuf_b, dom.tbuf_a)
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        for _ in range(3):
            dom.comm.all_to_all(dom.tbuf_b, dom.tbuf_a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)/3
        sent = dom.tbuf_a.numel()*8*(world - 1)/world
        probe = {'ms': round(ms, 3), 'bytes_sent_per_rank': int(sent),
                 'GBps_out_per_rank': round(sent/(ms*1e-3)/1e9, 1)}
    cnt = torch.tensor([parts.n, emig_total - emig0], dtype=torch.int64, device=dev)
    dist.all_reduce(cnt)
    # The check this line carries: the same step sequence once more, untimed,
    # with the identifiers travelling, and its id-keyed sample against the 1-rank values
    verify = None
    if not args.no_verify and not args.weak and fused and not args.dry_links:
        import numpy as np
        del parts
        torch.cuda.empty_cache()
        steps_total = args.warmup + args.steps
        rep = verify_replay(torch, args, dom, name, n_p, N, L, dev, rank, world, steps_total)
        tot = torch.tensor([float(rep['n_local']), rep['sum_mom2']], dtype=torch.float64,
                           device=dev)
        dist.all_reduce(tot)
        k = torch.tensor([len(rep['ids'])], dtype=torch.int64, device=dev)
        ks = [torch.empty_like(k) for _ in range(world)]
        dist.all_gather(ks, k)
        kmax = int(max(int(v) for v in ks))
        row = torch.zeros((kmax, 7), dtype=torch.float64, device=dev)
        if len(rep['ids']):
            row[:len(rep['ids']), 0] = torch.as_tensor(rep['ids'].astype(np.float64), device=dev)
            row[:len(rep['ids']), 1:4] = torch.as_tensor(rep['pos'], device=dev)
            row[:len(rep['ids']), 4:7] = torch.as_tensor(rep['mom'], device=dev)
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        if rank == 0:
            got = np.concatenate([r.cpu().numpy()[:int(c)] for r, c in zip(rows, ks)])
            got = got[np.argsort(got[:, 0])]
            key = verify_key(args, name, steps_total)
            timed_m2 = float(seen[:, 1].sum())
            verify = {
                'key': key, 'particles': int(tot[0].item()), 'particles_expected': n_p,
                'sum_mom2_replay': float(tot[1].item()), 'sum_mom2_timed_run': timed_m2,
                'timed_vs_replay_rel': abs(timed_m2 - float(tot[1].item()))/float(tot[1].item()),
                'sample': int(got.shape[0]),
                'what': ('the timed step sequence (tile sort, then warmup + steps x deposit / '
                         'solve / kick + drift + sort) run once more, untimed, on the same '
                         '(seed, identifier)-defined box with the identifiers travelling; the '
                         f'{VERIFY_SAMPLES} particles with identifiers k * n/{VERIFY_SAMPLES} '
                         'compared with the values a 1-rank run of this command left')}
            ref = verify_load(key)
            if ref is None:
                verify.update(reference=None, ok=None,
                              note='no 1-rank values for this key: run the same command with '
                                   '--gpus 1 first (it writes .bench_verify/<key>.json)')
            else:
                same = got.shape[0] == ref['ids'].shape[0] and \
                    np.array_equal(got[:, 0].astype(np.int64), ref['ids'])
                if same:
                    dx = np.abs(got[:, 1:4] - ref['pos'])
                    dx = np.minimum(dx, L - dx)
                    rms = np.sqrt((ref['mom']**2).mean()) or 1.0
                    dm = np.abs(got[:, 4:7] - ref['mom'])
                    verify.update(max_pos_err_over_boxsize=float(dx.max()/L),
                                  max_mom_err_over_rms=float(dm.max()/rms))
                verify.update(
                    reference=ref['path'], reference_made_by=ref['made_by'],
                    sum_mom2_reference=ref['sum_mom2'],
                    sum_mom2_rel_err=abs(float(tot[1].item()) - ref['sum_mom2'])/ref['sum_mom2'],
                    ok=bool(same and int(tot[0].item()) == ref['particles'] == n_p
                            and dx.max() <= 1e-12*L and dm.max() <= 1e-12*rms
                            and verify['timed_vs_replay_rel'] <= 1e-12))
    # shut the communicator down and push out what C stdio still holds (it went to stderr, see
    # main()), then give stdout back: the JSON line is the only line on it
    dist.destroy_process_group()
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if _saved_stdout is not None:
        os.dup2(_saved_stdout, 1)
    if rank != 0:
        return
    total, emigrants = int(cnt[0].item()), int(cnt[1].item())
    n_pl, n_gl = total/world, N**3/world
    mv = moved_bytes(n_pl, n_gl)
    # local kernels of rank 0: the gather-kick stage is one launch (+ the ghost fill messages)
    gk_name = 'kick_drift_sort' if fused else 'gather_kick'
    gk_ms = stages.get(gk_name, 0.0)
    gk_rate = mv[gk_name]/(gk_ms*1e-3)/1e9 if gk_ms else 0.0
    # transport: each transpose sends (P-1)/P of the local slab's transform, one peer per link
    tr_bytes = dom.tbuf_a.numel()*8*(world - 1)/world if world > 1 else 0
    ps_ms = stages.get('poisson+transposes+halos', 0.0)
    link_peak = min(world - 1, 7)*XGMI_LINK_GBS_DIR
    transport = None
    if world > 1:
        transport = {
            'bound': 'xgmi', 'unit': 'GB/s',
            'bytes_out_per_rank_per_transpose': int(tr_bytes), 'transposes_per_step': 2,
            'stage_ms': round(ps_ms, 3),
            'achieved': round(2*tr_bytes/(ps_ms*1e-3)/1e9, 1) if ps_ms else None,
            'peak': round(link_peak, 1),
            'frac': round(2*tr_bytes/(ps_ms*1e-3)/1e9/link_peak, 4) if ps_ms else None,
            'note': ('achieved = bytes one rank sends in the two FFT transposes / its whole '
                     'poisson+transposes stage (z/y/x transforms run under the links); peak = '
                     f'{min(world - 1, 7)} xGMI links x {XGMI_LINK_GBS_DIR} GB/s one way'
                     + ('' if backend == 'nccl' else
                        '; THIS RUN staged the messages through host memory (gloo): the '
                        'fraction says nothing about xGMI'))}
    # The link model of DESIGN.md §6 beside the measured stages, so that a scaling run can be
    # read stage by stage: local kernels at 1/P of their single-GPU times (measured on one
    # MI355X at this workload, profiles/r03_bench_ns_full_default.json), each transpose at
    # N^3*8/P^2 B per peer over that peer's own xGMI link, the z / y transforms hidden under
    # the links when those are slower (the transposes are pipelined with them), the x pass not.
    link_model = None
    if world > 1:
        link_model = link_model_predict(N, total, world, dom.tbuf_a.numel()*8)
        link_model.update(measured_step_ms=round(elapsed/args.steps*1e3, 3),
                          measured_poisson_stage_ms=round(ps_ms, 3))
    print(json.dumps({
        'metric': 'PM particle-updates/sec', 'value': total*args.steps/elapsed,
        'unit': 'particle-updates/s', 'steps_per_sec': args.steps/elapsed, 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed/args.steps*1e3,
        'timed_region_s': round(elapsed, 4),
        'higher_is_better': True, 'scaling': 'weak' if args.weak else 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'{name}: {total} particles (uniform random, thermal rms '
                               f'displacement {args.thermal} cells/step) / {N}^3 PM mesh, CIC, '
                               f'deconvolution order 4, FD order 2, {world} x-slab domains; '
                               '1 PM step = drift + exchange + tile sort + long-range kick'
                               + (' (kick, drift and sort in one pass)' if fused else ''),
                   'particles': total, 'gridsize': N, 'parallelism': f'xslab{world}',
                   'backend': ('rccl' if backend == 'nccl' else
                               f'{backend} (ranks share GPUs, host-staged messages: a functional '
                               'run of the sharded path, not a rate to quote)')},
        'emigrants_per_step': emigrants/args.steps,
        'emigrant_fraction_per_step': emigrants/args.steps/max(total, 1),
        'ranks_seen': {'world': world, 'backend': backend,
                       'rank_device': [[int(r[2]), int(r[3])] for r in seen]},
        'particles_per_rank': {'counts': [int(r[0]) for r in seen],
                               'max_over_mean': round(float(seen[:, 0].max()/seen[:, 0].mean()), 4)},
        'verify': verify,
        'roofline': {'bound': 'hbm', 'kernel': gk_name + ' (rank 0)',
                     'achieved': round(gk_rate, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(gk_rate/HBM_PEAK_GBS, 4), 'traffic': None,
                     'algorithmic_bytes': int(mv[gk_name]), 'kernel_ms': round(gk_ms, 4)},
        'transport': transport, 'link_model': link_model, 'cpu_baseline': None,
        'dry_links': dry, 'per_gpu': {'particle_updates_per_s': total*args.steps/elapsed/world,
                                      'particles': total//world, 'mesh_cells': N**3//world,
                                      'fft_backend': 'hand-written passes' if N & (N - 1) == 0
                                      else 'rocFFT per slab (grid size not a power of two)'},
        'stages_ms_rank0': {k: round(v, 3) for k, v in stages.items()},
        'transpose_probe_rank0': probe,
    }))


# ---------------------------------------------------------------------------
# self-spawn
# ---------------------------------------------------------------------------
_saved_stdout = None


def spawn_ranks(args):
    """--gpus N with no launcher environment: start the N ranks under torch.distributed.run
    (one per GPU, rendezvous on 127.0.0.1) and pass their output through; rank 0's JSON line is
    the last line of stdout."""
    import torch
    ngpu = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if ngpu < args.gpus:
        env['CONCEPT_BENCH_BACKEND'] = 'gloo'
        print(f'[bench] {args.gpus} ranks on {ngpu} visible GPU(s): ranks share GPUs and '
              'exchange through host memory (gloo)', file=sys.stderr, flush=True)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port',
           str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-baseline-child':
        return cpu_baseline_child(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default=None)
    ap.add_argument('--link-model', action='store_true',
                    help='print the link model\'s prediction for the headline size on 1, 2, 4 '
                         'and 8 GPUs (no GPU needed) and exit')
    ap.add_argument('--rung-loop', default=None, choices=['uniform', 'clustered'],
                    help='only the P3M time loop with 8 rungs at 256^3 / 512^3 '
                         '(configs.c2_p3m_rungs of the default line), --steps base steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true',
                    help='default run only: skip the other configurations timed after the '
                         'north-star region (the `configs` block of the JSON line)')
    ap.add_argument('--split-poisson', action='store_true',
                    help='time FFT forward / k-space kernel / FFT backward separately (unfused)')
    ap.add_argument('--weak', action='store_true',
                    help='weak scaling: 2^25 particles and ~1.3e8 mesh cells PER GPU '
                         '(512^3 / 640^3 / 800^3 / 1024^3 meshes on 1 / 2 / 4 / 8 GPUs) instead '
                         'of the fixed 2^28 / 1024^3 total')
    ap.add_argument('--dry-links', type=float, nargs='?', const=XGMI_LINK_GBS_DIR, default=0.0,
                    metavar='GB/s',
                    help='N > 1: replace every FFT transpose by a device sleep of (bytes to one '
                         'peer) / rate (default: one xGMI link, one way) — the pipelining '
                         'schedule and its overlap measured without a second GPU; results are '
                         'not meaningful')
    ap.add_argument('--no-verify', action='store_true',
                    help='skip the untimed replay with identifiers (the `verify` block)')
    ap.add_argument('--seed', type=int, default=1,
                    help='seed of the synthetic particles (SURVEY.md §8d: 1, 2, 3)')
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'lattice', 'clustered', 'zeldovich'],
                    help="particle distribution (SURVEY.md §8d): uniform random (U), displaced "
                         "lattice (Z: rms displacement 1.5 cells), or Gaussian blobs")
    ap.add_argument('--thermal', type=float, default=0.2,
                    help='rms displacement per step, in mesh cells, of the Maxwellian momenta '
                         'the particles start with (0: particles at rest)')
    ap.add_argument('--p3m', action='store_true',
                    help='P3M step (BASELINE configs[2]): long-range mesh with Gaussian cut-off + '
                         'short-range tile sweep (r_s = 1.25 cells, range 4.5 r_s, spline '
                         'softening 0.025*L/cbrt(N))')
    ap.add_argument('--no-fused', action='store_true',
                    help='PM: separate gather-kick and drift + sort kernels (the round-1 step) '
                         'instead of the fused kick + drift + scatter pass')
    ap.add_argument('--no-prepare', action='store_true',
                    help='do not fuse the next drift\'s tile histogram into the gather-kick')
    ap.add_argument('--no-sort', action='store_true',
                    help='direct (untiled) kernels on unsorted particles, for A/B')
    args = ap.parse_args()

    if args.link_model:
        n_p, N = WORKLOADS[args.workload or 'ns_256M_1024']
        rows = {str(P): link_model_predict(N, n_p, P) for P in (1, 2, 4, 8)}
        one = rows['1']['predicted_step_ms']
        for P, r in rows.items():
            r['predicted_speedup'] = round(one/r['predicted_step_ms'], 3)
            r['predicted_efficiency'] = round(one/r['predicted_step_ms']/int(P), 3)
        print(json.dumps({'workload': args.workload or 'ns_256M_1024', 'particles': n_p,
                          'gridsize': N, 'xgmi_GBps_per_link_one_way': XGMI_LINK_GBS_DIR,
                          'by_gpus': rows}, indent=1))
        return
    if args.dry_links:
        os.environ['CONCEPT_GPU_DRY_LINKS'] = str(args.dry_links)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    # CONCEPT_BENCH_BACKEND=gloo: ranks share the visible GPUs and exchange through host
    # memory — exercises the multi-rank code path on a 1-GPU box, never a reported rate
    backend = os.environ.get('CONCEPT_BENCH_BACKEND', 'nccl')
    ngpu = torch.cuda.device_count()
    if backend == 'nccl' and world > ngpu:
        backend = 'gloo'
    if backend != 'nccl':
        local_rank = local_rank % max(ngpu, 1)
    torch.cuda.set_device(local_rank)
    # CONCEPT_BENCH_FORCE_DIST=1: run the sharded code path (RCCL init, collectives) with a
    # single domain too — a smoke test of the N>1 path on a 1-GPU box
    force_dist = os.environ.get('CONCEPT_BENCH_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        # RCCL writes a version banner to stdout through C stdio: while the ranks run, file
        # descriptor 1 points at stderr; rank 0 gets it back for its one JSON line
        sys.stdout.flush()
        global _saved_stdout
        _saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        # a collective that a peer never joins fails after 5 minutes instead of hanging the box
        import datetime
        os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '1')
        limit = datetime.timedelta(minutes=5)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank),
                                    timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)

    if args.rung_loop:
        n_p, N = WORKLOADS[args.workload or 'c2_256c_512']
        print(json.dumps(rung_loop_leg(torch, torch.device('cuda', local_rank), args.rung_loop,
                                       base_steps=max(args.steps, 3), n_side=round(n_p**(1/3)),
                                       N=N)))
        return
    if args.workload in C4_SIZES:
        if world > 1:
            sys.exit('bench.py: the c4 workloads run on one GPU (tests/test_gpu_distributed.py '
                     'steps the shape over domains)')
        result = run_c4(args, torch, torch.device('cuda', local_rank), C4_SIZES[args.workload])
        print(json.dumps(result))
        return
    name = args.workload or ('c2_256c_512' if args.p3m else 'ns_256M_1024')
    if args.weak:
        if world not in WEAK:
            sys.exit(f'bench.py --weak: 1, 2, 4 or 8 GPUs (got {world})')
        name = f'weak_{world}x2^25'
        WORKLOADS[name] = WEAK[world]
    n_p, N = WORKLOADS[name]
    L = float(N)  # boxsize in grid units (synthetic)
    dev = torch.device('cuda', local_rank)
    if world > 1 or force_dist:
        return main_distributed(args, name, n_p, N, L, dev, rank, world, backend)
    result = run_single(args, torch, dev, rank)
    torch.cuda.empty_cache()
    if args.workload is None and not args.no_extra_configs and not args.p3m and not args.weak \
            and args.dist == 'uniform' and not (args.no_fused or args.no_sort or args.no_prepare
                                                or args.split_poisson):
        # The other configurations under the same clock: after the timed
        # north-star region, 20 steps each of BASELINE configs[1] as a PM and as a P3M step and
        # of the north-star size on the clustered, the displaced-lattice and the Zel'dovich-like
        #  distributions and with the seeds 2 and 3.  The headline fields
        # above are untouched.
        import copy
        result['configs'] = {}
        for cname, over in (('c2_256c_512_pm', dict(workload='c2_256c_512')),
                            ('c2_256c_512_p3m', dict(workload='c2_256c_512', p3m=True)),
                            ('c2_256c_512_p3m_clustered', dict(workload='c2_256c_512', p3m=True,
                                                               dist='clustered')),
                            ('ns_256M_1024_clustered', dict(dist='clustered')),
                            ('ns_256M_1024_lattice', dict(dist='lattice')),
                            ('ns_256M_1024_zeldovich', dict(dist='zeldovich')),
                            ('ns_256M_1024_seed2', dict(seed=2)),
                            ('ns_256M_1024_seed3', dict(seed=3))):
            a2 = copy.copy(args)
            a2.steps, a2.warmup = 20, 3
            for k, v in over.items():
                setattr(a2, k, v)
            r2 = run_single(a2, torch, dev, rank)
            torch.cuda.empty_cache()
            result['configs'][cname] = {
                'ms_per_step': round(r2['ms_per_step'], 4), 'steps': a2.steps,
                'particle_updates_per_s': r2['value'],
                'dominant_kernel': r2['roofline']['kernel'],
                'dominant_kernel_ms': r2['roofline']['kernel_ms'],
                'bound': r2['roofline']['bound'], 'frac': r2['roofline']['frac'],
                'phases_ms': {k: v['ms'] for k, v in r2['phases'].items()},
                'workload': r2['config']['workload']}
            if r2['roofline'].get('receivers_in_dense_tiles') is not None:
                result['configs'][cname]['receivers_in_dense_tiles'] = \
                    r2['roofline']['receivers_in_dense_tiles']
        r4 = run_c4(args, torch, dev, 512, steps=5, warmup=2)
        result['configs']['c4_nonlinnu_1gpu'] = {
            'ms_per_step': round(r4['ms_per_step'], 4), 'steps': r4['steps'],
            'particle_updates_per_s': r4['value'], 'dominant_kernel': r4['dominant_phase'],
            'dominant_kernel_ms': r4['roofline']['kernel_ms'], 'bound': r4['roofline']['bound'],
            'frac': r4['roofline']['frac'], 'interactions': r4['interactions'],
            'phases': r4['phases'], 'phases_ms': {k: v['ms'] for k, v in r4['phases'].items()},
            'workload': r4['config']['workload']}
        # BASELINE configs[3] at its own size on one GPU: 1024^3 particles /
        # 2048^3 mesh — 3 steps
        a3 = copy.copy(args)
        a3.steps, a3.warmup, a3.workload = 3, 1, 'c3_1024c_2048'
        r3 = run_single(a3, torch, dev, rank)
        torch.cuda.empty_cache()
        result['configs']['c3_1024c_2048'] = {
            'ms_per_step': round(r3['ms_per_step'], 4), 'steps': a3.steps,
            'particle_updates_per_s': r3['value'],
            'dominant_kernel': r3['roofline']['kernel'],
            'dominant_kernel_ms': r3['roofline']['kernel_ms'],
            'bound': r3['roofline']['bound'], 'frac': r3['roofline']['frac'],
            'phases_ms': {k: v['ms'] for k, v in r3['phases'].items()},
            'workload': r3['config']['workload']}
        # the reference's default run mode: the P3M time loop with 8 rungs
        result['configs']['c2_p3m_rungs'] = {
            'uniform': rung_loop_leg(torch, dev, 'uniform', base_steps=12),
            'clustered': rung_loop_leg(torch, dev, 'clustered', base_steps=10)}
        # the same workload through the drop-in API
        result['timeloop'] = timeloop_leg(torch, dev, result['ms_per_step'])
        result['timeloop_ms_per_step'] = result['timeloop']['ms_per_base_step']
    if not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(name)
    print(json.dumps(result))


C4_SIZES = {'c4_nonlinnu_1gpu': 512, 'c4_nonlinnu_small': 128, 'c4_nonlinnu_tiny': 32}


def c4_components(torch, dev, size, seed=1, thermal=0.2, mass=1.0, dt=1e-4):
    """The components of BASELINE configs[4]'s shape (param/example_nonlinnu:36-45 with
    _size = size): size^3 matter particles on P3M (mesh 2 size), a fluid with non-linear energy and
    momentum density on a (size/2)^3 grid, global PM grid size/2; fixed step integrals.
    Returns (params, particles, fluid, ᔑdt, ᔑdt_rungs, the particles' P3M mesh)."""
    import numpy as np
    from concept_amd import commons
    from concept_amd.mesh import get_mesh
    from concept_amd.species import Component
    n_side, N3, N1 = size, 2*size, size//2
    n = n_side**3
    L = float(N3)
    p = commons.load_params({
        'boxsize': L,
        'potential_options': {'gridsize': {'global': {'gravity': {'pm': N1, 'p3m': N3}}}},
        'select_forces': {'particles': {'gravity': 'p3m'}, 'fluid': {'gravity': 'pm'}},
        'select_softening_length': {'particles': '0.025*boxsize/cbrt(N)'}})
    part = Component('matter', 'matter', N=n, mass=mass)
    gen = torch.Generator(device=dev).manual_seed(seed)
    torch.rand((n, 3), dtype=torch.float64, device=dev, generator=gen, out=part.pos)
    part.pos.mul_(L*(1 - 1e-13))
    torch.randn((n, 3), dtype=torch.float64, device=dev, generator=gen, out=part.mom)
    part.mom.mul_(thermal/3**0.5*(L/N3)*mass/dt)
    fluid = Component('neutrino', 'matter', gridsize=N1, boltzmann_order=1)
    # a smooth density contrast of 10 % and a matching momentum density
    x = (torch.arange(N1, dtype=torch.float64, device=dev) + 0.5)*(2*np.pi/N1)
    wave = torch.sin(x)[:, None, None]*torch.cos(2*x)[None, :, None]*torch.sin(3*x)[None, None, :]
    mean = 0.02*n*mass/N1**3   # a few per cent of the matter's mass in the fluid
    fluid.ϱ.copy_(mean*(1 + 0.1*wave))
    fluid.𝒫.copy_(1e-3*fluid.ϱ)
    for d in range(3):
        fluid.J[d].copy_(1e-2*mean*wave)
    del x, wave
    sdt = {'1': dt, 'a**(-2)': dt}
    for c in (part, fluid):
        sdt['a**(-3*w_eff)', c.name] = dt
        sdt['a**(-3*w_eff-1)', c.name] = dt
    # (one integral per rung index, main.py:1203-1215; every particle sits on rung 0)
    sdt_rungs = {('a**(-3*w_eff₀-3*w_eff₁-1)', part.name, part.name):
                 np.full(3*getattr(part, 'N_rungs', 8) - 1, dt)}
    mesh3 = get_mesh(N3, L, p.nghosts, p.cell_centered, 2, dev)
    return p, part, fluid, sdt, sdt_rungs, mesh3


def run_c4(args, torch, dev, size=512, steps=None, warmup=None):
    """BASELINE configs[4]'s shape on ONE GPU: param/example_nonlinnu:36-45
    with _size = 512 — 512^3 matter particles (P3M on a 1024^3 mesh, spline-softened short range
    with the default r_s = 1.25 cells, range 4.5 r_s) + a fluid component with non-linear energy
    and momentum density on a 256^3 grid (the neutrino's place; species 'matter' here, the
    reference's test/fluid_gravity stand-in, CLASS being out of reach), global PM grid 256.
    find_interactions() makes three mesh solves of a long kick: (p3m: particles <- particles) on
    1024^3, (pm: particles <- fluid) and (pm: fluid <- particles, fluid) on 256^3
    (interactions.py:2456-2636).  A step here is the loop body of main.timeloop() for that
    configuration, driven through the drop-in API itself — Component.drift_sort, gravity(...,
    'short-range') + apply_Δmom, then gravity(..., 'long-range') per interaction — with fixed
    step integrals; the fluid's own evolution (fluid.py) is outside the path.  Phases are timed
    by HIP events around each call; `moved` bytes per phase below."""
    from concept_amd import interactions
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    p, part, fluid, sdt, sdt_rungs, mesh3 = c4_components(torch, dev, size, args.seed, args.thermal)
    comps = [part, fluid]
    n, N3, N1, L, dt = part.N, 2*size, size//2, p.boxsize, sdt['1']
    long_range = interactions.find_interactions(comps, 'long-range')
    short_range = interactions.find_interactions(comps, 'short-range')
    describe = lambda it: (f"{it.method}: {', '.join(c.name for c in it.receivers)} <- "
                           f"{', '.join(c.name for c in it.suppliers)}")
    PH = ['drift_sort', 'short_range'] + ['long: ' + describe(it) for it in long_range]
    part.tile_sort(mesh=mesh3)
    events = []

    def step(record):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(PH) + 1)] if record else None
        k = 0

        def mark():
            nonlocal k
            if ev is not None:
                ev[k].record()
                k += 1
        mark()
        part.drift_sort(sdt, mesh=mesh3)
        mark()
        part.nullify_Δ('mom')
        for it in short_range:
            getattr(interactions, it.force)(it.method, it.receivers, it.suppliers, sdt_rungs,
                                            'short-range', False)
        part.apply_Δmom()
        mark()
        for it in long_range:
            getattr(interactions, it.force)(it.method, it.receivers, it.suppliers, sdt,
                                            'long-range', False)
            mark()
        if record:
            events.append(ev)
    for _ in range(warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mesh3.check_errors()
    ms = {ph: sum(ev[k].elapsed_time(ev[k + 1]) for ev in events)/len(events)
          for k, ph in enumerate(PH)}
    # bytes each phase must move: N_p particles, G3 = 1024^3-type mesh of the
    # P3M interaction, G1 = 256^3-type mesh of the PM interactions and the fluid's grids
    Np, G3, G1 = n, N3**3, N1**3
    # (drift_sort: pos + mom read and written, and the Component's other columns — identifiers,
    # populated order, Δmom, the two rung arrays — read and written at their new places)
    moved = {'drift_sort': 96*Np + (16 + 16 + 48 + 4)*Np,
             'short_range': 2*24*Np + 8*Np + 72*Np + 72*Np}       # cell list, sweep, Δmom apply
    for ph, it in zip(PH[2:], long_range):
        rec_p = [c for c in it.receivers if c.representation == 'particles']
        rec_f = [c for c in it.receivers if c.representation == 'fluid']
        sup_p = [c for c in it.suppliers if c.representation == 'particles']
        sup_f = [c for c in it.suppliers if c.representation == 'fluid']
        G = G3 if it.method == 'p3m' else G1
        b = 80*G                                                  # five in-place passes
        b += len(sup_p)*(24*Np + 8*G) + len(sup_f)*16*G           # deposit; ϱ read, mesh written
        if sup_p and sup_f:
            b += 32*G                                             # a second forward transform
        b += len(rec_p)*(72*Np + 8*G)                             # gather-kick
        b += len(rec_f)*3*(8*G + 16*G + 16*G)                     # fluid kick per dimension:
        #                                           potential, J read + written, ϱ and 𝒫 read
        moved[ph] = b
    phases = {ph: {'ms': round(ms[ph], 4), 'moved_GB': round(moved[ph]/1e9, 3),
                   'GBps': round(moved[ph]/(ms[ph]*1e-3)/1e9, 1),
                   'frac_hbm': round(moved[ph]/(ms[ph]*1e-3)/1e9/HBM_PEAK_GBS, 4)}
              for ph in PH}
    # The short-range phase is the pair sweep (FP64 issue bound) with the cell list in front of
    # it: its fraction of the HBM rate says nothing, executed pair tests per second against the
    # FP64 issue rate do (counted by one more pass through the short-range interactions after
    # the timed region, on a scratch Δmom; `kernel_ms` is the whole phase, list included)
    del phases['short_range']['frac_hbm']
    part.nullify_Δ('mom')
    mesh3.shortrange_stats(True)
    for it in short_range:
        getattr(interactions, it.force)(it.method, it.receivers, it.suppliers, sdt_rungs,
                                        'short-range', False)
    sweep_roofline = pair_sweep_roofline(mesh3.shortrange_stats(False), ms['short_range'],
                                         'short_range')
    part.nullify_Δ('mom')
    phases['short_range'].update(
        bound='valu_fp64', frac_valu_fp64=sweep_roofline['frac'],
        tests_per_hit=sweep_roofline['tests_per_hit'], lane_use=sweep_roofline['lane_use'])
    ms_per_step = elapsed/steps*1e3
    dom = max(PH, key=lambda ph: ms[ph])
    out = {
        'metric': 'P3M + PM (particles + fluid) particle-updates/sec', 'value': n*steps/elapsed,
        'unit': 'particle-updates/s', 'steps_per_sec': steps/elapsed, 'n_gpus': 1, 'steps': steps,
        'warmup': warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'c4 (param/example_nonlinnu with _size = {size}): {n} matter '
                               f'particles, P3M mesh {N3}^3, fluid grid and global PM grid '
                               f'{N1}^3, three mesh solves per long kick; through '
                               'Component.drift_sort / gravity() / apply_Δmom',
                   'particles': n, 'gridsize': N3, 'fluid_gridsize': N1,
                   'parallelism': 'domains1'},
        'interactions': [describe(it) for it in long_range],
        'phases': phases, 'dominant_phase': dom,
        'roofline': ({'bound': 'hbm', 'kernel': dom, 'achieved': phases[dom]['GBps'],
                      'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': phases[dom]['frac_hbm'],
                      'traffic': None, 'kernel_ms': phases[dom]['ms']}
                     if dom != 'short_range' else sweep_roofline),
    }
    del part, fluid, comps
    from concept_amd import mesh as mesh_module
    mesh_module.free_meshes()
    torch.cuda.empty_cache()
    return out


def timeloop_leg(torch, dev, raw_ms, n=2**28, N=1024, base_steps=24):
    """Twice: a Component as the reference's own — no identifiers, particle order free
    (keep_order = False: no 64-bit column rides through the passes) — which is the headline
    figure, and one whose host() restores the populated order (the default of this API: one
    identifier column travels)."""
    out = timeloop_run(torch, dev, raw_ms, n, N, base_steps, keep_order=False)
    ordered = timeloop_run(torch, dev, raw_ms, n, N, base_steps, keep_order=True)
    out['with_order_column'] = {k: ordered[k] for k in (
        'ms_per_base_step', 'ratio_to_raw_step', 'ms_per_synchronisation_step',
        'mean_ms_over_all_steps', 'one_pass_steps', 'particles_kept')}
    # both contracts side by side: the API's default (host() restores the populated
    # order) and the reference's own (no identifiers, order free), which README.md quotes
    out['ms_per_base_step_default'] = ordered['ms_per_base_step']
    out['ms_per_base_step_no_order'] = out['ms_per_base_step']
    return out


def rung_loop_leg(torch, dev, dist, base_steps=12, n_side=256, N=512):
    """The reference's default run mode under the bench's clock: BASELINE
    configs[2]'s size — 256^3 particles, P3M on a 512^3 mesh — through stepper.Timeloop with
    N_rungs = 8: a white-noise field at rest (`uniform`) or bench.py's clustered box from
    a = 0.02, every base step the rung loop of driftkick_short (main.py:1347-1624: a drift, the
    short-range kick of the active rungs and the rung bookkeeping per sub-step) + the long-range
    kick.  Wall time between the beginnings of consecutive base steps; the loop's calls grouped
    by HIP events around them."""
    import statistics
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from concept_amd import commons, shortrange, species, stepper
    from concept_amd.mesh import PotentialMesh
    from tools.sr_positions import positions
    n = n_side**3
    p = commons.load_params({
        'boxsize': float(N), 'H0': 0.07, 'Ωb': 0.05, 'Ωcdm': 0.25, 'a_begin': 0.02,
        'output_times': {'a': (0.5,)},
        'potential_options': {'gridsize': {'gravity': {'p3m': N}}},
        'select_forces': {'all': {'gravity': 'p3m'}}})
    c = species.Component('matter', 'matter', N=n, mass=p.ρ_mbar*p.boxsize**3/n)
    gen = torch.Generator(device=dev).manual_seed(13)
    c.pos.copy_(positions(dist, n, p.boxsize, gen))
    c.mom.zero_()
    # the loop's calls, grouped: an event pair around each (and the host's clock at both ends)
    groups, host = {}, []

    def timed(obj, attr, group):
        f = getattr(obj, attr)

        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            try:
                return f(*a, **k)
            finally:
                e1.record()
                groups.setdefault(group, []).append((e0, e1))
                host.append((group, t0, time.perf_counter()))
        setattr(obj, attr, wrapper)
        return lambda: setattr(obj, attr, f)
    undo = [timed(species.Component, 'substep_finish', 'wait_for_the_sub_step'),
            timed(species.Component, 'substep_begin', 'drift_flag_nullify'),
            timed(species.Component, 'substep_end', 'apply_convert_jumps_populations'),
            timed(PotentialMesh, 'shortrange_cells', 'cell_list'),
            timed(PotentialMesh, 'shortrange_sweep_cells', 'sweep'),
            timed(PotentialMesh, 'shortrange_sparse', 'sweep_without_list'),
            timed(stepper.RungStepper, 'kick_long', 'long_range_kick')]

    class Enough(Exception):
        pass
    stamps, marks, sparse0 = [], [], shortrange.sparse_sweeps

    def on_step(lp):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        marks.append({g: len(v) for g, v in groups.items()})
        if len(stamps) > base_steps + 2:
            raise Enough
    loop = stepper.Timeloop([c], on_step=on_step)
    try:
        loop.run()
    except Enough:
        pass
    torch.cuda.synchronize()
    for u in undo:
        u()
    # the first two base steps out (the loop's set-up, the lists' first allocations)
    dts = [(b - a)*1e3 for a, b in zip(stamps[2:-1], stamps[3:])]
    lo = marks[2]
    per_group = {g: round(sum(e0.elapsed_time(e1) for e0, e1 in v[lo.get(g, 0):])/len(dts), 3)
                 for g, v in groups.items()}
    per_group.pop('wait_for_the_sub_step', None)
    # the host between the end of a sub-step's wait and the first call that launches work for
    # the next one (the GPU has nothing queued then), per base step
    t_first = stamps[2]
    idle, inside = 0.0, {}
    prev = None
    for g, t0, t1 in host:
        if t0 < t_first:
            prev = (g, t1)
            continue
        inside[g] = inside.get(g, 0.0) + (t1 - t0)
        if prev is not None and prev[0] == 'wait_for_the_sub_step' and t0 <= stamps[-1]:
            idle += t0 - prev[1]
        prev = (g, t1)
    sweeps = len(groups.get('sweep', [])) - lo.get('sweep', 0)
    out = {'ms_per_base_step': round(statistics.median(dts), 3),
           'ms_per_base_step_min_max': [round(min(dts), 3), round(max(dts), 3)],
           'base_steps_timed': len(dts),
           'sub_steps_per_base_step': round(
               (len(groups.get('drift_flag_nullify', [])) - lo.get('drift_flag_nullify', 0))/len(dts), 2),
           'sweeps_per_base_step': round(sweeps/len(dts), 2),
           'gpu_ms_per_base_step_by_call': per_group,
           'gpu_ms_per_base_step_in_calls': round(sum(per_group.values()), 3),
           'host_ms_per_base_step_between_wait_and_next_call': round(idle*1e3/len(dts), 3),
           'host_ms_per_base_step_inside_calls': {g: round(v*1e3/len(dts), 3)
                                                  for g, v in inside.items()},
           'rung_populations_at_the_end': [int(v) for v in c.rungs_N],
           'a_reached': loop.cosmo.a, 'particles_kept': int(c.N_local) == n,
           'what': (f'stepper.Timeloop (= main.timeloop(), main.py:102-471), P3M, N_rungs = '
                    f'{p.N_rungs}, {n_side}^3 particles / {N}^3 mesh, {dist} box at rest from '
                    'a = 0.02; median wall time between the beginnings of consecutive base '
                    'steps (torch.cuda.synchronize() in the step callback); by_call = GPU time '
                    'between HIP events around the calls of the loop, per base step')}
    del c, loop
    from concept_amd import mesh as mesh_module
    mesh_module.free_meshes()
    torch.cuda.empty_cache()
    return out


def timeloop_run(torch, dev, raw_ms, n, N, base_steps, keep_order):
    """The metric's workload through the API the boundary promises instead of through the raw
    kernel sequence: concept_amd.stepper.Timeloop = main.timeloop() (main.py:102-471) on a
    Component of 2^28 particles with a 1024^3 PM mesh and the matter + Λ clock from a = 0.1 —
    background, time-step integrals, limiters (v_rms after every kick), the streaming form of
    the loop (every long kick riding with the drift after it: deposit, solve, one fused pass
    per base step).  Wall time between the beginnings of consecutive base steps, host included;
    a synchronisation step (every 8th, and at reductions of Δt) takes two passes."""
    import statistics
    from concept_amd import commons, stepper
    from concept_amd.species import Component
    p = commons.load_params({
        'boxsize': float(N), 'H0': 0.07, 'Ωb': 0.05, 'Ωcdm': 0.25, 'a_begin': 0.1,
        'output_times': {'a': (0.5,)},
        'potential_options': {'gridsize': {'gravity': {'pm': N}}},
        'select_forces': {'all': {'gravity': 'pm'}}})
    mass = p.ρ_mbar*p.boxsize**3/n
    c = Component('matter', 'matter', N=n, mass=mass)
    c.keep_order = keep_order
    gen = torch.Generator(device=dev).manual_seed(5)
    torch.rand((n, 3), dtype=torch.float64, device=dev, generator=gen, out=c.pos)
    c.pos.mul_(p.boxsize*(1 - 1e-13))
    u = 100*p.units.km/p.units.s   # peculiar velocities of ~100 km/s: mom = a m u
    torch.randn((n, 3), dtype=torch.float64, device=dev, generator=gen, out=c.mom)
    c.mom.mul_(0.1*mass*u/3**0.5)

    class Enough(Exception):
        pass
    stamps, passes = [], []

    def on_step(lp):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
        passes.append(lp.stream_passes)
        if len(stamps) > base_steps:
            raise Enough
    loop = stepper.Timeloop([c], on_step=on_step)
    replays = stepper.stream_replays
    try:
        loop.run()
    except Enough:
        pass
    torch.cuda.synchronize()
    dts = [(b - a)*1e3 for a, b in zip(stamps[:-1], stamps[1:])]
    dps = [b - a for a, b in zip(passes[:-1], passes[1:])]
    plain = [t for t, k in zip(dts, dps) if k == 1]   # base steps of one pass
    ms = statistics.median(plain) if plain else float('nan')
    out = {'ms_per_base_step': round(ms, 3), 'base_steps_timed': len(dts),
           'one_pass_steps': len(plain),
           'ms_per_synchronisation_step': (round(statistics.median(
               [t for t, k in zip(dts, dps) if k > 1]), 3) if len(plain) < len(dts) else None),
           'mean_ms_over_all_steps': round(sum(dts)/max(len(dts), 1), 3),
           'ratio_to_raw_step': round(ms/raw_ms, 4), 'raw_ms_per_step': round(raw_ms, 3),
           'stream_passes': loop.stream_passes, 'wrong_guesses': loop.stream_wrong_guesses,
           'replays': stepper.stream_replays - replays, 'particles_kept': int(c.N_local) == n,
           'a_reached': loop.cosmo.a, 'keep_order': keep_order,
           'what': ('stepper.Timeloop (= main.timeloop(), main.py:102-471) on a Component of 2^28 '
                    'particles / 1024^3 PM mesh, matter + Λ clock from a = 0.1, streaming form; '
                    'median wall time between the beginnings of consecutive one-pass base steps '
                    '(deposit + solve + fused kick/drift/sort with the sum of mom^2 for v_rms + '
                    'host), torch.cuda.synchronize() in the step callback')}
    del c, loop
    from concept_amd import mesh as mesh_module
    mesh_module.free_meshes()
    torch.cuda.empty_cache()
    return out


def pair_sweep_roofline(st, ms, kernel, cell_offsets=None, nt=None):
    """the short-range sweep's roofline entry from the counters of cg_shortrange_stats (`st`)
    and the sweep's time: EXECUTED pair tests per second against the FP64 vector issue rate"""
    tests = st['cells'][0] + st['dense'][0]
    hits = st['cells'][1] + st['dense'][1]
    trips = st['cells'][2] + st['dense'][2]
    per_test = 10   # 3 sub, 1 mul + 2 fma (r2), 1 cmp, 3 fma
    dense_tiles = dense_receivers = None
    if cell_offsets is not None:
        # the receivers the dense tiles' sweep took: populations of the tiles from the cell list
        nc = 2*nt
        off = cell_offsets.long()
        pop = (off[1:] - off[:-1]).reshape(nc//2, 2, nc//2, 2, nc//2, 2).sum((1, 3, 5))
        dense_min = int(os.environ.get('CONCEPT_GPU_SR_DENSE_MIN', '64'))
        took = st['dense'][0] > 0 and dense_min > 0
        dense_tiles = int((pop >= dense_min).sum()) if took else 0
        dense_receivers = int(pop[pop >= dense_min].sum()) if took else 0
    # 256 CUs x 4 SIMDs x 16 FP64 lanes/clk x 2.4 GHz: one FP64 VALU op per lane slot
    valu_peak = 256*4*16*2.4e9
    return {
        'bound': 'valu_fp64', 'kernel': kernel, 'unit': 'pair-tests/s',
        'achieved': round(tests/(ms*1e-3), 1), 'peak': round(valu_peak/per_test, 1),
        'frac': round(tests/(ms*1e-3)/(valu_peak/per_test), 4), 'traffic': None,
        'kernel_ms': round(ms, 4), 'pair_tests_per_launch': int(tests),
        'pairs_in_range_per_launch': int(hits),
        'tests_per_hit': round(tests/max(hits, 1), 3),
        'lane_slots_per_launch': int(64*trips),
        'lane_use': round(tests/max(64*trips, 1), 4),
        'by_kernel': {k: {'pair_tests': v[0], 'in_range': v[1], 'wave_trips': v[2]}
                      for k, v in st.items()},
        'dense_tiles': dense_tiles, 'receivers_in_dense_tiles': dense_receivers,
        'note': ('the sweep is not HBM-bound (72 B per particle against hundreds of pair '
                 f'tests); peak = FP64 vector issue rate {valu_peak:.3g} lane-ops/s / '
                 f'{per_test} FP64 VALU instructions per pair test; pair tests = EXECUTED '
                 'tests counted on the device (half-tile cells sweep + dense tiles\' sweep), '
                 'tests_per_hit = executed tests per pair inside the force range, lane_use = '
                 'tests per lane slot of the pair loops')}


def run_single(args, torch, dev, rank=0):
    """The timed steps of one configuration on one GPU: the result dict of the JSON line
    (without cpu_baseline)."""
    from concept_amd.mesh import PotentialMesh
    name = args.workload or ('c2_256c_512' if args.p3m else 'ns_256M_1024')
    if args.weak:
        name = 'weak_1x2^25'
        WORKLOADS[name] = WEAK[1]
    n_p, N = WORKLOADS[name]
    L = float(N)  # boxsize in grid units (synthetic)
    mesh = PotentialMesh(N, L, nghosts=2)
    # step scalars: fixed (enable_Hubble=False semantics)
    mass = 1.0
    G = 1.0
    dt = 1e-4
    if args.dist == 'uniform':
        # a function of (seed, identifier): the box an N-rank run of this command steps too
        pos, mom, _ = global_particles(torch, args, n_p, L, L/N, mass, dt, dev)
    else:
        gen = torch.Generator(device=dev).manual_seed(args.seed + rank)
        pos = make_positions(torch, args, n_p, N, L, dev, gen, mesh)
        mom = thermal_momenta(torch, args, pos.shape, L/N, mass, dt, dev, gen)
    fused = not (args.no_fused or args.no_sort or args.p3m or args.no_prepare)
    if fused:
        # the fused kick + drift + scatter keeps the particles in tile REGIONS WITH GAPS
        # (capacities predicted from the present populations): arrays of that capacity
        cap = mesh.region_capacity(n_p)
        grow = lambda t: torch.cat([t, torch.empty((cap - n_p, 3), dtype=t.dtype, device=dev)])
        pos, mom = grow(pos), grow(mom)
    pos2, mom2 = torch.empty_like(pos), torch.empty_like(mom)
    table = mesh.new_tile_table()
    contribution = (dt/dt)*mass*(float(N)**(-3)*(N/L)**3)
    C = -L**2*G/3.141592653589793
    kick_factor = mass*(-dt)
    dt_over_mass = dt/mass

    poisson = ['fft_forward', 'kspace', 'fft_backward'] if args.split_poisson else ['poisson']
    sr = None
    if args.p3m:
        from concept_amd import commons as _commons
        from concept_amd import shortrange as _sr
        scale = 1.25*L/N
        rng_ = 4.5*scale
        nt_sr = int((L/1)/rng_*(1 + _commons.machine_ϵ))
        soft = 0.025*L/round(n_p**(1/3))
        sr_table, maxr2 = _sr.get_shortrange_table(soft, scale, rng_, 4096, 'spline', dev)
        dmom = torch.zeros_like(mom)
        sr = dict(scale=scale, range=rng_, nt=nt_sr, table=sr_table, scaling=4095/maxr2,
                  r2_max=rng_**2, factor=G*mass*mass*dt, E=-(2*3.141592653589793/L*scale)**2)
    if fused:
        # one sort into tile order up front (untimed, like generating the particles); every
        # timed step is deposit -> solve -> kick + drift + sort: the same cycle of one drift,
        # one sort, one deposit, one solve and one kick per step, entered after the sort
        PHASES = ['deposit'] + poisson + ['kick_drift_sort']
        mesh.sort_particles(pos[:n_p], mom[:n_p], None, pos2[:n_p], mom2[:n_p], None, table)
        pos, pos2, mom, mom2 = pos2, pos, mom2, mom
        reg = {'start': table[:mesh.table_entries], 'count': None,
               'spare': mesh.new_region_table()}
        spare2 = mesh.new_region_table()
    elif args.no_sort:
        PHASES = ['drift', 'zero', 'deposit'] + poisson + ['gather_kick']
    else:  # the tiled deposit assigns the mesh: no zero-fill pass
        PHASES = ['drift_sort', 'deposit'] + poisson + ['gather_kick']
    if args.p3m:
        PHASES += ['sr_cells', 'sr_sweep']
    events = []

    def step(record):
        nonlocal pos, mom, pos2, mom2
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(PHASES) + 1)] \
            if record else None
        i = 0

        def mark():
            nonlocal i
            if ev is not None:
                ev[i].record()
                i += 1
        mark()
        if fused:
            if reg['count'] is None:
                mesh.deposit_tiled(pos[:n_p], table, contribution, accumulate=False)
            else:
                mesh.deposit_regions(pos, reg['start'], reg['count'], contribution)
            mark()
            if args.split_poisson:
                mesh.poisson_forward(4, C, False, 0.0, apply_kernel=False)
                mark()
                mesh.poisson_kernel(4, C, False, 0.0)
                mark()
                mesh.poisson_backward()
            else:
                mesh.poisson_solve(4, C, False, 0.0)
            mark()
            start_out, count_out = reg['spare']
            mesh.predict_regions(reg['start'], reg['count'], start_out)
            mesh.gather_kick_drift_scatter(pos, mom, None, reg['start'], reg['count'], pos2, mom2,
                                           None, start_out, count_out, 2, kick_factor,
                                           dt_over_mass)
            mark()
            old = (reg['start'], reg['count']) if reg['count'] is not None else spare2
            reg.update(start=start_out, count=count_out, spare=old)
            pos, pos2, mom, mom2 = pos2, pos, mom2, mom
            if record:
                events.append(ev)
            return
        if not args.no_sort:
            # drift + tile sort fused (cg_drift_sort): the drifted particles land in tile order
            mesh.drift_sort(pos, mom, None, pos2, mom2, None, dt_over_mass, table)
            pos, pos2 = pos2, pos
            mom, mom2 = mom2, mom
            mark()
            mesh.deposit_tiled(pos, table, contribution, accumulate=False)
            mark()
        else:
            mesh.drift(pos, mom, dt_over_mass)
            mark()
            mesh.zero()
            mark()
            mesh.deposit(pos, contribution)
            mark()
        if args.split_poisson:
            mesh.poisson_forward(4, C, False, 0.0, apply_kernel=False)
            mark()
            mesh.poisson_kernel(4, C, False, 0.0)
            mark()
            mesh.poisson_backward()
            mark()
        else:  # the product path: k-space kernel fused into the x pass of the FFT
            mesh.poisson_solve(4, C, sr is not None, sr['E'] if sr else 0.0)
            mark()
        order = 4 if sr else 2  # differentiation defaults: pm 2, p3m 4 (commons.py:3209-3237)
        if not args.no_sort and not sr and not args.no_prepare:
            # the long kick is the last momentum update before the next drift: histogram the
            # drifted tile keys here, the next cg_drift_sort skips its first pass
            mesh.gather_kick_tiled_prepare(pos, mom, table, order, kick_factor, dt_over_mass)
        elif not args.no_sort:
            mesh.gather_kick_tiled(pos, mom, table, order, kick_factor)
        else:
            mesh.gather_kick(pos, mom, order, kick_factor)
        mark()
        if sr:
            cells = sr['cells'] = mesh.shortrange_cells(pos, sr['nt'], L/sr['nt'])
            mark()
            # kick_short (main.py:1173-1262): nullify Δmom, sweep, apply — the apply is
            # fused into the sweep's store: the target is the momentum array itself
            mesh.shortrange_sweep_cells(cells, mom, cells, sr['nt'], sr['table'],
                                        sr['scaling'], sr['r2_max'], sr['factor'])
            mark()
        if record:
            events.append(ev)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mesh.check_errors()   # (a bucket that outgrew its region would have dropped particles)
    timed_m2 = None
    if fused:
        kept = int(reg['count'].long().sum().item())
        if kept != n_p:
            sys.exit(f'bench.py: {kept} of {n_p} particles after the timed steps')
        timed_m2 = mesh.measure_momentum_regions(mom, reg['start'], reg['count'])[0]

    n_g = N**3
    mv, sv = moved_bytes(n_p, n_g), survey_bytes(n_p, n_g)
    phase_ms = {ph: 0.0 for ph in PHASES}
    for ev in events:
        for k, ph in enumerate(PHASES):
            phase_ms[ph] += ev[k].elapsed_time(ev[k + 1])
    for ph in PHASES:
        phase_ms[ph] /= len(events)

    # every timed step by itself (HIP events: a step's first mark to the next step's; the last
    # one to its own last mark): a cold first configuration shows as such
    step_ms = [events[i][0].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]
    step_ms.append(events[-1][0].elapsed_time(events[-1][len(PHASES)]))

    def entry(key, ms):
        gbs = mv[key]/(ms*1e-3)/1e9 if ms > 0 else 0.0
        e = {'ms': round(ms, 4), 'moved_GB': round(mv[key]/1e9, 3), 'GBps': round(gbs, 1),
             'frac_hbm': round(gbs/HBM_PEAK_GBS, 4)}
        if key in sv and sv[key] != mv[key]:
            e['survey_8d_GB'] = round(sv[key]/1e9, 3)  # the unfused reference phases' bytes
        return e
    phases = {ph: entry(ph, phase_ms[ph]) for ph in PHASES}
    # single kernels: the phases that are one launch, plus the five FFT passes timed by
    # HIP events inside the library (a few extra solves after the timed region)
    kernels = {ph: phase_ms[ph] for ph in ('deposit', 'gather_kick', 'kick_drift_sort', 'drift',
                                           'sr_sweep')
               if ph in phase_ms}  # (drift_sort is two kernels + a scan: reported under phases)
    if not args.split_poisson:
        pass_ms = [0.0]*5
        reps = 3
        lr, E = (True, sr['E']) if sr else (False, 0.0)
        for _ in range(reps):
            for k, v in enumerate(mesh.poisson_solve_timed(4, C, lr, E)):
                pass_ms[k] += v/reps
        names = ('fft_z_forward', 'fft_y_forward', 'fft_x_fused_kspace', 'fft_y_backward',
                 'fft_z_backward')
        if pass_ms[1] < 0.05*pass_ms[0]:
            # the z and y passes run interleaved over cache-sized chunks of layers (cg_fft.hip):
            # they are timed together; the second pass of a chunk is served by the infinity
            # cache (its bytes still cross L2 <-> fabric, which is what `moved` counts)
            names = ('fft_zy_forward_chunked', 'fft_x_fused_kspace', 'fft_yz_backward_chunked')
            pass_ms = [pass_ms[0] + pass_ms[1], pass_ms[2], pass_ms[3] + pass_ms[4]]
        for nm, v in zip(names, pass_ms):
            kernels[nm] = v
    dom = max(kernels, key=lambda k: kernels[k])
    ms_per_step = elapsed/args.steps*1e3
    result = {
        'metric': ('P3M' if args.p3m else 'PM') + ' particle-updates/sec',
        'value': n_p*args.steps/elapsed,
        'unit': 'particle-updates/s', 'steps_per_sec': args.steps/elapsed,
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'timed_region_s': round(elapsed, 4),
        'ms_per_step_min_median_max': [round(min(step_ms), 4),
                                       round(sorted(step_ms)[len(step_ms)//2], 4),
                                       round(max(step_ms), 4)],
        'higher_is_better': True, 'scaling': 'weak' if args.weak else 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'{name}: {n_p} particles ({args.dist}, seed {args.seed}, thermal rms '
                               f'displacement {args.thermal} cells/step) / {N}^3 PM mesh, CIC, '
                               f'deconvolution order 4, FD order {4 if sr else 2}, 1 '
                               + ('P3M step = drift + tile sort + long-range kick + short-range '
                                  'kick (r_s 1.25 cells, range 4.5 r_s)' if sr else
                                  'PM step = drift + tile sort + long-range kick'
                                  + (' (kick, drift and sort of a step in one pass: the loop is '
                                     'entered after the sort)' if fused else '')),
                   'particles': n_p, 'gridsize': N, 'parallelism': 'domains1'},
        'emigrants_per_step': 0,
    }
    if dom == 'sr_sweep':
        # FP64 VALU-bound: EXECUTED pair tests/s against the FP64 vector issue rate.  The tests
        # are counted by the sweeps themselves (cg_shortrange_stats: counting instantiations of
        # the two kernels, one extra sweep over the last step's cell list after the timed
        # region, on a scratch copy of the momenta): a lane that holds a receiver against a
        # supplier of its range is one test, a test with r2 <= r2_max one hit.
        scratch = mom.clone()
        mesh.shortrange_stats(True)
        mesh.shortrange_sweep_cells(sr['cells'], scratch, sr['cells'], sr['nt'], sr['table'],
                                    sr['scaling'], sr['r2_max'], sr['factor'])
        st = mesh.shortrange_stats(False)
        del scratch
        result['roofline'] = pair_sweep_roofline(st, kernels[dom], dom, sr['cells'][1], sr['nt'])
    else:
        ach = mv[dom]/(kernels[dom]*1e-3)/1e9
        traffic, traffic_source = pmc_traffic(dom, name)
        result['roofline'] = {
            'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': round(ach/HBM_PEAK_GBS, 4), 'traffic': traffic,
            'algorithmic_bytes': mv[dom], 'kernel_ms': round(kernels[dom], 4),
            'hbm_rate_GBps': (round(traffic/(kernels[dom]*1e-3)/1e9, 1) if traffic else None),
            'survey_8d_bytes': sv.get(dom),
            'note': ('algorithmic bytes = what this fused kernel must move (DESIGN.md §4); '
                     'survey_8d_bytes = SURVEY.md §8(d)\'s figure for the unfused reference '
                     'phases it replaces, given for comparison only'),
            'traffic_source': traffic_source}
    result['kernels'] = {k: entry(k, ms) for k, ms in kernels.items()}
    interp = 'kick_drift_sort' if fused else 'gather_kick'
    groups = {'deposit+interp': (['deposit', interp], phase_ms['deposit'] + phase_ms[interp]),
              'poisson_solve': (['poisson'] if not args.split_poisson else
                                ['fft_forward', 'kspace', 'fft_backward'],
                                sum(phase_ms[ph] for ph in poisson))}
    result['roofline_groups'] = {}
    for gname, (keys, ms) in groups.items():
        moved = sum(mv[k] for k in keys)
        credit = (sv['deposit'] + sv[interp]) if gname == 'deposit+interp' else sv['poisson']
        result['roofline_groups'][gname] = {
            'ms': round(ms, 3), 'moved_GB': round(moved/1e9, 2),
            'frac_moved': round(moved/(ms*1e-3)/1e9/HBM_PEAK_GBS, 4),
            # SURVEY.md §8(d)'s bytes for the unfused reference phases of the group: a credit
            # for work removed by fusion, stated as bytes only — never turned into a fraction
            'survey_8d_GB': round(credit/1e9, 2)}
    result['phases'] = phases
    del pos, mom, pos2, mom2
    if fused and args.dist == 'uniform' and not args.no_verify and not args.weak \
            and n_p <= 2**29:   # (the replay keeps the identifiers in two stores: ~300 B per particle)
        # the values an N-rank run of this command is checked against (its `verify` block):
        # the same step sequence once more, untimed, with the identifiers travelling
        import types
        torch.cuda.empty_cache()
        steps_total = args.warmup + args.steps
        rep = verify_replay(torch, args, types.SimpleNamespace(mesh=mesh), name, n_p, N, L, dev,
                            0, 1, steps_total)
        import numpy as np
        order = np.argsort(rep['ids'])
        key = verify_key(args, name, steps_total)
        path = verify_save(key, rep['ids'][order], rep['pos'][order], rep['mom'][order],
                           rep['n_local'], rep['sum_mom2'],
                           {'made_by': f'bench.py --gpus 1 --steps {args.steps} --warmup '
                                       f'{args.warmup} --seed {args.seed}'
                                       + (f' --workload {args.workload}' if args.workload else '')})
        result['verify'] = {
            'key': key, 'role': 'reference (1 rank)', 'written_to': os.path.relpath(path, REPO),
            'particles': rep['n_local'], 'particles_expected': n_p,
            'sum_mom2_replay': rep['sum_mom2'], 'sum_mom2_timed_run': timed_m2,
            'timed_vs_replay_rel': abs(timed_m2 - rep['sum_mom2'])/rep['sum_mom2'],
            'sample': int(len(order)),
            'ok': bool(rep['n_local'] == n_p
                       and abs(timed_m2 - rep['sum_mom2']) <= 1e-12*rep['sum_mom2']),
            'what': ('the timed step sequence run once more, untimed, through the sharded code '
                     'path (ParticleStore / RegionParticles / pm_step_regions on one domain) with '
                     f'the identifiers travelling; the {VERIFY_SAMPLES} particles with '
                     f'identifiers k * n/{VERIFY_SAMPLES} are what `--gpus N` runs of this '
                     'command compare their own with')}
    mesh.close()
    return result


if __name__ == '__main__':
    main()
