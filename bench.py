#!/usr/bin/env python3
"""bench.py — PM steps/s and particle-updates/s of the MI355X gravity stepper.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" = one full PM base step of the reference's time loop
(main.py:335-358: drift -> long-range kick) on synthetic, HBM-resident
particles: drift (A11) + tile sort, mesh zero + CIC deposit (A1/A2), rocFFT
R2C (A4), k-space Poisson kernel (A5/A6), C2R (A8), fused finite-difference +
CIC gather + kick (A9/A10).  Workload at N=1: BASELINE.json's metric
configuration, 2^28 (~256M) particles on a 1024^3 mesh, FP64.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant hand-written
kernel, from HIP events on the stream the kernels run on; `cpu_baseline` is
the C oracle (oracle/, a port of the reference's algorithm) timed on a
bounded sample on this host's cores.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec

WORKLOADS = {
    # name: (particles, gridsize)
    'ns_256M_1024': (2**28, 1024),       # BASELINE.json metric config
    'c2_256c_512': (256**3, 512),        # configs[1]
    'c1_128c_256': (128**3, 256),        # configs[0]
    'c3_1024c_2048': (1024**3, 2048),    # configs[3] (meant for 8 GPUs; fits one: ~175 GB)
    'tiny': (32**3, 64),
}


def algorithmic_bytes(n_p, n_g):
    """SURVEY.md §8(d) per-phase algorithmic bytes (FP64), adjusted to what
    this build's fused kernels must move at minimum (stated in DESIGN.md)."""
    return {
        'zero': 8*n_g,
        'deposit': 24*n_p + 8*n_g,               # read pos, write (accumulate) grid
        'fft_forward': 48*n_g,                   # 3 passes x (read + write)
        'kspace': 16*n_g,
        'fft_backward': 48*n_g,
        'poisson': 48*n_g + 16*n_g + 48*n_g,     # A4 + A5-A7 + A8
        'sr_cells': 2*24*n_p + 8*n_p,            # pos read twice, order written
        'sr_sweep': 24*n_p + 48*n_p,             # not HBM-bound: FP64/LDS pair arithmetic
        'gather_kick': 24*n_p + 48*n_p + 8*n_g,  # pos, mom RMW, potential once (FD fused)
        'drift': 48*n_p + 24*n_p,
        'drift_sort': (48*n_p + 24*n_p) + (2*(48*n_p) + 24*n_p),  # drift row + sort row
        'sort': 2*(48*n_p) + 24*n_p,             # histogram reads pos; scatter moves pos+mom
    }


def cpu_baseline():
    """The oracle (a C port of the reference's algorithm) on bounded samples of the workload,
    timed twice: with OpenMP over the particle and plane loops + scipy's threaded pocketfft
    (the reference runs one MPI rank per core) at the best of a few thread counts, and on
    one core with numpy's pocketfft (the reference's own pure-Python FFT).  The threaded
    figure is the reported baseline.  Thread counts: on the 256-thread bench host 16-32
    threads are fastest (9.6 M particle-updates/s; 128 threads: 5.3 M — the deposit's atomic
    adds and the FFT do not scale further; tools/cpu_baseline_probe.py)."""
    import numpy as np
    from oracle import oracle
    oracle.build()

    def run(fast, sample_n, sample_grid, nsteps):
        n, L = sample_n**3, float(sample_grid)
        rng = np.random.default_rng(7)
        pos = rng.uniform(0, L, (n, 3))
        mom = np.zeros((n, 3))
        t0 = time.perf_counter()
        for _ in range(nsteps):
            oracle.drift(pos, mom, 1e-3, L, fast=fast)
            oracle.pm_long_range(pos, mom, mass=1.0, boxsize=L, gridsize=sample_grid,
                                 G_Newton=1.0, dt_1=1e-3, dt_dens=1e-3, dt_kick=1e-3,
                                 diff_order=2, fast=fast, want_indices=False)
        return time.perf_counter() - t0

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    omp = oracle.lib('omp')
    trial = {}
    for threads in sorted({min(avail, t) for t in (16, 32)}):
        omp.orc_threads(threads)
        run('omp', 128, 256, 1)  # thread pool, page faults
        trial[threads] = run('omp', 128, 256, 2)
    cores = min(trial, key=trial.get)
    omp.orc_threads(cores)
    steps_all, steps_one = 8, 12
    dt_all = run('omp', 256, 512, steps_all)
    dt_one = run(True, 128, 256, steps_one)
    flags = '-O3 -funroll-loops -ffast-math (reference src/Makefile flags)'
    return {
        'value': 256**3*steps_all/dt_all, 'unit': 'particle-updates/s', 'cores': cores,
        'kind': 'port', 'steps_per_sec': steps_all/dt_all,
        'sample': f'256^3 particles / 512^3 mesh (BASELINE configs[1] size), {steps_all} PM '
                  f'steps, oracle C port built {flags} -fopenmp on {cores} of {avail} host '
                  f'threads (fastest of {sorted(trial)}) + scipy.fft with {cores} workers, '
                  f'{dt_all:.1f} s wall',
        'single_thread': {
            'value': 128**3*steps_one/dt_one, 'unit': 'particle-updates/s', 'cores': 1,
            'steps_per_sec': steps_one/dt_one,
            'sample': f'128^3 particles / 256^3 mesh (BASELINE configs[0]), {steps_one} PM '
                      f'steps, oracle C port built {flags} + numpy pocketfft, {dt_one:.1f} s wall'},
    }


def main_distributed(args, name, n_p, N, L, dev, rank, world):
    """Strong scaling: the same total workload on `world` x-slab domains, one per GPU
    (concept_amd/distributed.py).  A step = drift + particle exchange + tile sort (fused:
    DistributedParticles.drift_exchange_sort) + long-range kick (deposit, ghost fold, FFT
    with two all-to-all transposes, ghost fill, gather-kick)."""
    import torch
    import torch.distributed as dist
    from concept_amd.distributed import DistributedParticles, SlabDomain, pm_kick
    dom = SlabDomain(N, L, device=dev)
    n_local = n_p//world
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    pos = torch.rand((n_local, 3), dtype=torch.float64, device=dev, generator=gen)
    # uniform inside this rank's slab: lower CIC cell x in [x0, x0 + nxl)  <=>
    # x in [(x0 + 1/2) cells, (x0 + nxl + 1/2) cells), wrapped into the box
    cell = L/N
    pos[:, 0] = torch.remainder((dom.mesh.x0 + 0.5 + pos[:, 0]*dom.nxl*(1 - 1e-12))*cell, L)
    pos[:, 1:] *= L
    pos.clamp_(min=0.0, max=float(torch.nextafter(torch.tensor(L, dtype=torch.float64),
                                                  torch.tensor(0.0, dtype=torch.float64))))
    parts = DistributedParticles(dom, pos, torch.zeros_like(pos), None, slack=1.15)
    del pos
    parts.exchange()
    parts.tile_sort()
    mass, G, dt = 1.0, 1.0, 1e-4
    contribution = mass*(float(N)**(-3)*(N/L)**3)
    C = -L**2*G/3.141592653589793

    stage_events = []  # per timed step: [(name, event), ...] on the compute stream

    def step(record=False):
        # fused: emigrants of the coming drift are shipped first, then one drift + sort pass
        # pair; the gather-kick histograms the tiles of the next drift
        evs = []

        def mark(name):
            if record:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append((name, e))
        mark('start')
        parts.drift_exchange_sort(dt/mass)
        mark('drift+exchange+sort')
        pm_kick(dom, parts, contribution, 4, C, mass*(-dt), diff_order=2,
                next_dt_over_mass=dt/mass, mark=mark)
        if record:
            stage_events.append(evs)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    # this rank's stage times (the exchanges wait for the slowest peer inside their stage)
    stages = {}
    for evs in stage_events:
        for (_, a), (name, b) in zip(evs[:-1], evs[1:]):
            stages[name] = stages.get(name, 0.0) + a.elapsed_time(b)/len(stage_events)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
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
    cnt = torch.tensor([parts.n], dtype=torch.int64, device=dev)
    dist.all_reduce(cnt)
    # RCCL prints its version banner through C stdio; push it out (and shut the communicator
    # down) before the result so that the JSON line is the last line of stdout
    dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if rank != 0:
        return
    total = int(cnt.item())
    print(json.dumps({
        'metric': 'PM particle-updates/sec', 'value': total*args.steps/elapsed,
        'unit': 'particle-updates/s', 'steps_per_sec': args.steps/elapsed, 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed/args.steps*1e3,
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
        'data': 'synthetic',
        'config': {'workload': f'{name}: {total} particles (uniform random) / {N}^3 PM mesh, '
                               f'CIC, deconvolution order 4, FD order 2, {world} x-slab domains; '
                               '1 PM step = drift + exchange + tile sort + long-range kick',
                   'particles': total, 'gridsize': N, 'parallelism': f'xslab{world}'},
        'roofline': None, 'cpu_baseline': None,
        'stages_ms_rank0': {k: round(v, 3) for k, v in stages.items()},
        'transpose_probe_rank0': probe,
    }))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--split-poisson', action='store_true',
                    help='time FFT forward / k-space kernel / FFT backward separately (unfused)')
    ap.add_argument('--dist', default='uniform', choices=['uniform', 'lattice', 'clustered'],
                    help="particle distribution (SURVEY.md §8d): uniform random (U), displaced "
                         "lattice (Z: rms displacement 1.5 cells), or Gaussian blobs")
    ap.add_argument('--p3m', action='store_true',
                    help='P3M step (BASELINE configs[2]): long-range mesh with Gaussian cut-off + '
                         'short-range tile sweep (r_s = 1.25 cells, range 4.5 r_s, spline '
                         'softening 0.025*L/cbrt(N))')
    ap.add_argument('--no-prepare', action='store_true',
                    help='do not fuse the next drift\'s tile histogram into the gather-kick')
    ap.add_argument('--no-sort', action='store_true',
                    help='direct (untiled) kernels on unsorted particles, for A/B')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    # CONCEPT_BENCH_BACKEND=gloo: ranks share cuda:0 and exchange through host memory —
    # only to exercise the multi-rank code path on a 1-GPU box, never a reported number
    backend = os.environ.get('CONCEPT_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # CONCEPT_BENCH_FORCE_DIST=1: run the sharded code path (RCCL init, collectives) with a
    # single domain too — a smoke test of the N>1 path on a 1-GPU box
    force_dist = os.environ.get('CONCEPT_BENCH_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from concept_amd.mesh import PotentialMesh
    name = args.workload or ('c2_256c_512' if args.p3m else 'ns_256M_1024')
    n_p, N = WORKLOADS[name]
    L = float(N)  # boxsize in grid units (synthetic)
    dev = torch.device('cuda', local_rank)
    if world > 1 or force_dist:
        return main_distributed(args, name, n_p, N, L, dev, rank, world)
    mesh = PotentialMesh(N, L, nghosts=2)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    pos = torch.rand((n_p, 3), dtype=torch.float64, device=dev, generator=gen)
    if args.dist == 'uniform':
        pos.mul_(L)
    elif args.dist == 'lattice':
        side = round(n_p**(1/3))
        if side**3 != n_p:
            sys.exit('--dist lattice needs a cubic particle count (e.g. --workload c2_256c_512)')
        idx = torch.arange(n_p, device=dev)
        lat = torch.stack([idx//(side*side), (idx//side) % side, idx % side], 1).double()
        disp = torch.randn((n_p, 3), dtype=torch.float64, device=dev, generator=gen)*1.5*(L/N)
        pos = torch.remainder((lat + 0.5)*(L/side) + disp, L)
        del idx, lat, disp
    else:  # clustered: 64 Gaussian blobs of sigma = L/40 holding 80 % of the particles
        centres = torch.rand((64, 3), dtype=torch.float64, device=dev, generator=gen)*L
        which = torch.randint(0, 64, (n_p,), device=dev, generator=gen)
        blob = centres[which] + torch.randn((n_p, 3), dtype=torch.float64, device=dev,
                                            generator=gen)*(L/40)
        keep = torch.rand(n_p, dtype=torch.float64, device=dev, generator=gen) < 0.2
        pos = torch.where(keep[:, None], pos*L, torch.remainder(blob, L))
        del centres, which, blob, keep
    pos.clamp_(min=0.0, max=float(torch.nextafter(torch.tensor(L, dtype=torch.float64),
                                                  torch.tensor(0.0, dtype=torch.float64))))
    mom = torch.zeros_like(pos)
    pos2, mom2 = torch.empty_like(pos), torch.empty_like(mom)
    table = mesh.new_tile_table()
    # step scalars: fixed (enable_Hubble=False semantics)
    mass = 1.0
    G = 1.0
    dt = 1e-4
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
    if args.no_sort:
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
            dmom.zero_()
            cells = mesh.shortrange_build(pos, sr['nt'], L/sr['nt'])
            mark()
            mesh.shortrange_sweep(pos, cells, dmom, pos, cells, sr['nt'], True, sr['table'],
                                  sr['scaling'], sr['r2_max'], sr['factor'])
            mom.add_(dmom)
            mark()
        if record:
            events.append(ev)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        return
    n_g = N**3
    alg = algorithmic_bytes(n_p, n_g)
    phase_ms = {ph: 0.0 for ph in PHASES}
    for ev in events:
        for k, ph in enumerate(PHASES):
            phase_ms[ph] += ev[k].elapsed_time(ev[k + 1])
    for ph in PHASES:
        phase_ms[ph] /= len(events)
    phases = {}
    for ph in PHASES:
        gbs = alg[ph]/(phase_ms[ph]*1e-3)/1e9 if phase_ms[ph] > 0 else 0.0
        phases[ph] = {'ms': round(phase_ms[ph], 4), 'alg_GB': round(alg[ph]/1e9, 3),
                      'GBps': round(gbs, 1), 'frac_hbm': round(gbs/HBM_PEAK_GBS, 4)}
    # single kernels: the phases that are one launch, plus the five FFT passes timed by
    # HIP events inside the library (a few extra solves after the timed region)
    kernels = {ph: (alg[ph], phase_ms[ph]) for ph in ('deposit', 'gather_kick', 'drift')
               if ph in phase_ms}  # (drift_sort is two kernels + a scan: reported under phases)
    if not args.split_poisson:
        pass_ms = [0.0]*5
        reps = 3
        for _ in range(reps):
            for k, v in enumerate(mesh.poisson_solve_timed(4, C, False, 0.0)):
                pass_ms[k] += v/reps
        # SURVEY.md §8(d) accounting: forward 48*N_g (3 passes), k-space 16*N_g, inverse
        # 48*N_g.  The fused x pass stands for the forward x pass + the k-space kernel +
        # the inverse x pass (16 + 16 + 16); the five entries sum to the 112*N_g of the row.
        names, bytes_ = (('fft_z_forward', 'fft_y_forward', 'fft_x_fused_kspace',
                          'fft_y_backward', 'fft_z_backward'), (16, 16, 48, 16, 16))
        if pass_ms[1] < 0.05*pass_ms[0]:
            # the z and y passes run interleaved over cache-sized chunks of layers (cg_fft.hip):
            # they are timed together; the second pass of a chunk is served by the infinity
            # cache, which is how the pair exceeds the HBM rate an isolated pass can reach
            names, bytes_ = (('fft_zy_forward_chunked', 'fft_x_fused_kspace',
                              'fft_yz_backward_chunked'), (32, 48, 32))
            pass_ms = [pass_ms[0] + pass_ms[1], pass_ms[2], pass_ms[3] + pass_ms[4]]
        for nm, v, b in zip(names, pass_ms, bytes_):
            kernels[nm] = (b*n_g, v)
    dom = max(kernels, key=lambda k: kernels[k][1])
    ach = kernels[dom][0]/(kernels[dom][1]*1e-3)/1e9
    traffic = None
    try:  # HBM bytes per launch from the committed PMC run of this same command
        pmc = json.load(open(os.path.join(REPO, 'profiles', 'r01_pmc_hbm_traffic.json')))
        key = {'fft_x_fused_kspace': 'k_fft_strided_p<10,512,2,8>',
               'gather_kick': 'k_gather_kick_tiled<2,16,true>', 'drift': 'k_drift',
               'deposit': 'k_deposit_cic_pull<16,false>',
               'fft_y_forward': 'k_fft_strided_p<10,512,0,8>',
               'fft_y_backward': 'k_fft_strided_p<10,512,1,8>',
               'fft_zy_forward_chunked': None, 'fft_yz_backward_chunked': None,
               'fft_z_forward': 'k_fft_z_forward<10,128>',
               'fft_z_backward': 'k_fft_z_backward<10,128>'}.get(dom)
        if name == 'ns_256M_1024' and key:
            traffic = pmc['kernels'][key]['total_GB']*1e9
    except Exception:
        traffic = None
    groups = {
        'deposit+interp': (alg['deposit'] + alg['gather_kick'],
                           phase_ms['deposit'] + phase_ms['gather_kick']),
        'poisson_solve': (alg['poisson'], sum(phase_ms[ph] for ph in poisson)),
    }
    ms_per_step = elapsed/args.steps*1e3
    result = {
        'metric': ('P3M' if args.p3m else 'PM') + ' particle-updates/sec',
        'value': n_p*world*args.steps/elapsed,
        'unit': 'particle-updates/s', 'steps_per_sec': args.steps/elapsed,
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'{name}: {n_p} particles ({args.dist}, seed 1) / {N}^3 PM mesh, '
                               'CIC, deconvolution order 4, FD order 2, 1 PM step = drift + '
                               'tile sort + long-range kick', 'particles': n_p, 'gridsize': N,
                   'parallelism': f'domains{world}'},
        'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1),
                     'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach/HBM_PEAK_GBS, 4),
                     'traffic': traffic, 'algorithmic_bytes': kernels[dom][0],
                     'kernel_ms': round(kernels[dom][1], 4),
                     'hbm_rate_GBps': (round(traffic/(kernels[dom][1]*1e-3)/1e9, 1)
                                       if traffic else None),
                     'note': ('algorithmic bytes follow SURVEY.md §8(d); for the fused x pass '
                              'they are those of the three reference passes it replaces '
                              '(x forward + k-space + x inverse = 48*N_g), its measured HBM '
                              'traffic is one read + one write of the mesh'
                              if dom == 'fft_x_fused_kspace' else
                              'algorithmic bytes are the fused kernel\'s minimum (DESIGN.md §4): '
                              'pos + mom read-modify-write + the potential once = 72*N_p + 8*N_g; '
                              'SURVEY.md §8(d) row A10 counts three force grids (72*N_p + '
                              '24*N_g), see unfused_accounting'
                              if dom == 'gather_kick' else
                              'algorithmic bytes follow SURVEY.md §8(d) / DESIGN.md §4'),
                     'traffic_source': 'profiles/r01_pmc_hbm_traffic.json (rocprofv3 --pmc '
                                       'FETCH_SIZE / WRITE_SIZE, separate passes)'},
        'kernels': {k: {'alg_GB': round(b/1e9, 3), 'ms': round(ms, 4),
                        'GBps': round(b/(ms*1e-3)/1e9, 1),
                        'frac_hbm': round(b/(ms*1e-3)/1e9/HBM_PEAK_GBS, 4)}
                    for k, (b, ms) in kernels.items()},
        'roofline_groups': {k: {'alg_GB': round(b/1e9, 2), 'ms': round(ms, 3),
                                'GBps': round(b/(ms*1e-3)/1e9, 1),
                                'frac': round(b/(ms*1e-3)/1e9/HBM_PEAK_GBS, 4)}
                            for k, (b, ms) in groups.items()},
        'phases': phases,
    }
    if dom == 'gather_kick':
        # the same launch priced with SURVEY.md §8(d)'s own A10 row (three force grids read)
        b = 72*n_p + 24*n_g
        a = b/(kernels[dom][1]*1e-3)/1e9
        result['roofline']['unfused_accounting'] = {
            'algorithmic_bytes': b, 'achieved': round(a, 1), 'frac': round(a/HBM_PEAK_GBS, 4)}
    if not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline()
    print(json.dumps(result))


if __name__ == '__main__':
    main()
