"""Multi-rank x-slab PM path against the single-domain path, on ONE GPU: P
processes share cuda:0 and exchange over gloo (staged through host memory).
The production transport is RCCL with one GPU per rank; the kernels, layouts
and exchange logic exercised here are the same.  Bar: the reference's own
nprocs-independence bar (test/nprocs_pm/analyze.py:121) is 1e-9*boxsize on
positions; we require 1e-12 of the rms kick / 1e-13*boxsize."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def single_domain(N, n_side, steps, p3m=False, cells_per_step=0.0):
    import torch
    from concept_amd import commons, shortrange
    from concept_amd.mesh import PotentialMesh
    L = 64.0
    mesh = PotentialMesh(N, L)
    rng = np.random.default_rng(77)
    n = n_side**3
    pos = torch.tensor(rng.uniform(0, L, (n, 3)), device='cuda')
    contribution, C, kick, dtm = 0.37, -2.5, -0.002, 0.9
    # (cells_per_step: rms displacement per step and dimension in mesh cells; 0: unit momenta)
    sigma = cells_per_step*(L/N)/dtm if cells_per_step else 1.0
    if cells_per_step:
        contribution *= 1.4e-3   # (kicks about half the size of the thermal momenta)
    mom = torch.tensor(rng.normal(0, sigma, (n, 3)), device='cuda')
    for step in range(steps):
        if p3m:
            scale = 1.25*L/N
            rng_ = 4.5*scale
            nt = int(L/rng_*(1 + commons.machine_ϵ))
            table, maxr2 = shortrange.get_shortrange_table(0.05*L/n_side, scale, rng_, 4096,
                                                           'spline', pos.device)
            cells = mesh.shortrange_cells(pos, nt, L/nt)
            dm = torch.zeros_like(mom)
            mesh.shortrange_sweep_cells(cells, dm, cells, nt, table, 4095/maxr2, rng_**2, 3e-4)
            mom += dm
        mesh.zero()
        mesh.deposit(pos, contribution)
        mesh.poisson_solve(4, C, p3m, -(2*np.pi/L*1.25*L/N)**2 if p3m else 0.0)
        mesh.gather_kick(pos, mom, 2 + 2*(step % 2), kick)
        mesh.drift(pos, mom, dtm)
    return pos.cpu().numpy(), mom.cpu().numpy()


@pytest.mark.parametrize('world,N,p3m', [(2, 32, False), (4, 64, False), (2, 64, False),
                                         (8, 128, False), (2, 64, True), (4, 128, True),
                                         (2, 32, 'fused'), (4, 64, 'fused'), (8, 128, 'fused'),
                                         (2, 32, 'regions'), (4, 64, 'regions'),
                                         (8, 128, 'regions'),
                                         # the production instantiation of the FFT passes (even /
                                         # odd split pass writing / reading the all-to-all
                                         # buffers blocked by destination domain), one step
                                         (2, 1024, False), (4, 1024, False)])
def test_slab_domains_match_single_domain(world, N, p3m, n_side=20, steps=None):
    """p3m = 'fused': the PM step with the fused drift + exchange + sort
    (DistributedParticles.drift_exchange_sort) and the tile histogram prepared by the
    gather-kick; 'regions': kick + drift + tile sort in one pass over particles kept in tile
    regions with gaps (RegionParticles) — what bench.py runs on N > 1 GPUs."""
    steps_given = steps
    steps = 3
    mode = p3m if isinstance(p3m, str) else ('p3m' if p3m else 'pm')
    p3m = p3m is True
    if mode in ('fused', 'regions'):
        steps = 5
    if N >= 1024:
        steps = 1
    if steps_given is not None:
        steps = steps_given
    cells_per_step = 0.4 if n_side > 20 else 0.0
    pos_ref, mom_ref = single_domain(N, n_side, steps, p3m, cells_per_step)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port))
            procs.append(subprocess.Popen(
                [sys.executable, os.path.join(REPO, 'tests', 'dist_worker.py'), tmp, str(N),
                 str(n_side), str(steps), 'gloo', mode, str(cells_per_step)], env=env,
                stdout=subprocess.PIPE,
                stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=1500)[0].decode() for p in procs]
        for r, p in enumerate(procs):
            assert p.returncode == 0, f'rank {r} failed:\n{outs[r][-3000:]}'
        ids, pos, mom = [], [], []
        for r in range(world):
            d = np.load(os.path.join(tmp, f'rank{r}.npz'))
            ids.append(d['ids'])
            pos.append(d['pos'])
            mom.append(d['mom'])
    ids = np.concatenate(ids)
    assert np.array_equal(np.sort(ids), np.arange(n_side**3))  # every particle exactly once
    pos_d = np.empty_like(pos_ref)
    mom_d = np.empty_like(mom_ref)
    pos_d[ids] = np.concatenate(pos)
    mom_d[ids] = np.concatenate(mom)
    # the bar is on the KICKS (the momenta themselves are of order one where the kicks are
    # 1e-3 of that: 1e-12 of their rms said little), plus the rounding of adding a kick to a
    # momentum, once per kick
    rng = np.random.default_rng(77)
    rng.uniform(0, 64.0, (n_side**3, 3))
    mom0 = rng.normal(0, cells_per_step*(64.0/N)/0.9 if cells_per_step else 1.0, (n_side**3, 3))
    kick_rms = np.sqrt(((mom_ref - mom0)**2).mean())
    assert kick_rms > 0
    # (1e-12 of the rms kick per kick, adding up as a random walk over the steps)
    assert np.abs(mom_d - mom_ref).max() <= 1e-12*kick_rms*max(1.0, steps**0.5) \
        + 2.3e-16*(2*steps)*np.abs(mom_ref).max()
    dx = np.abs(pos_d - pos_ref)
    dx = np.minimum(dx, 64.0 - dx)
    assert dx.max() <= 1e-13*64.0


@pytest.mark.parametrize('world,N,mode', [(2, 64, 'regions'), (4, 128, 'regions'), (2, 64, False)])
def test_slab_domains_with_the_tile_order(world, N, mode, monkeypatch):
    """The same with the tile kernels walking their tiles in cgk_tile_order's order on boxes this
    small too (CONCEPT_GPU_TILE_ORDER_MIN=-1) and every tile above 1.5 times the mean population
    counted as heavy: the slab deposit's ghost row behind the ordered tiles, the gather-kick and
    the fused pass of every domain."""
    monkeypatch.setenv('CONCEPT_GPU_TILE_ORDER_MIN', '-1')
    test_slab_domains_match_single_domain(world, N, mode)


def test_rccl_async_layer_exchange_and_pipelined_solve():
    """The production transport on the one GPU there is: a 1-rank RCCL group.  Checks that the
    asynchronous piecewise all_to_all (lists of views, work handles) is accepted by RCCL and
    that the chunk-pipelined distributed solve — forced on although a single rank would
    normally take the local path — equals the single-domain solve."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    code = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from concept_amd.distributed import Comm
comm = Comm()
assert not comm.stage
P, nxl, row = 1, 32, 1000
inp = torch.rand(P*nxl*row, dtype=torch.float64, device='cuda')
out = torch.full_like(inp, float('nan'))
works = [comm.all_to_all_layers(out, inp, nxl, l0, 8, async_op=True) for l0 in range(0, nxl, 8)]
for w in works:
    w.wait()
torch.cuda.synchronize()
assert torch.equal(out, inp)
# the pipelined transposing solve against the local one
from concept_amd.distributed import SlabDomain
from concept_amd.mesh import PotentialMesh
os.environ['CONCEPT_GPU_DIST_FORCE'] = '1'
# 64: the generic kernels; 1024: the production instantiation (even/odd split passes on the
# transposed buffer, cache-sized pieces)
for N in (64, 1024):
    L = float(N)
    gen = torch.Generator(device='cuda').manual_seed(N)
    rho = torch.rand((N, N, N), dtype=torch.float64, device='cuda', generator=gen)
    ref = PotentialMesh(N, L)
    dom = SlabDomain(N, L)
    assert len(dom.pieces) >= 4
    for m in (ref, dom.mesh):
        m.zero()
        m.fluid_add(rho, 1.0, '=')
    del rho
    ref.poisson_solve(4, -2.5)
    dom.poisson_solve(4, -2.5)
    torch.cuda.synchronize()
    pos = torch.rand((5000, 3), dtype=torch.float64, device='cuda')*L
    va = torch.zeros((5000, 3), dtype=torch.float64, device='cuda')
    vb = torch.zeros_like(va)
    ref.gather_kick(pos, va, 2, 1.0)
    dom.mesh.gather_kick(pos, vb, 2, 1.0)
    scale = va.abs().max().item()
    assert scale > 0 and (va - vb).abs().max().item() <= 1e-12*scale, \
        (N, (va - vb).abs().max().item()/scale)
    ref.close()
    dom.mesh.close()
    del ref, dom
dist.destroy_process_group()
print('RCCL-PIECES-OK')
''' % REPO
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'RCCL-PIECES-OK' in out, out[-3000:]


def _run_ranks(world, script, args, timeout=900):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', script)] + args,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=timeout)[0].decode() for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and f'RANK{r}-OK' in outs[r], f'rank {r} failed:\n{outs[r][-4000:]}'


# The Component / gravity() layer over domains: the bodies of the
# single-domain golden tests, unchanged, on 2 and 4 ranks.  gravity(), Component.drift(),
# stepper.timeloop and RungStepper are collective; results are gathered by Component.host().
COMPONENT_CASES = [
    ('pm_api', 'pm_n16_g32'),               # gravity('pm') + drift + tile_sort + gravity again
    ('steps', 'steps_pm_n8_g16'),           # A18: init half kick, drift -> kick, PM
    ('steps', 'steps_p3m_n8_g32'),          # ... P3M: long + short kicks, boundary suppliers
    ('rungs', '-'),                         # adaptive rungs incl. jumps (rungs_p3m_n8_g32)
    ('p3m_kick', 'p3m_n8_g32'),
    ('k1', '1'), ('k1', '2'), ('k1', '5'),  # lattice stays put: pairs across components, domains
    ('mixed', 'fluid_pm_n8_g16'),           # particles + fluid on the shared mesh
    ('nonlinnu', '-'),                      # configs[4]'s shape: three mesh solves, two grid sizes
    ('multigrid', 'multigrid_n8_pow2'),     # copy_modes between grids: rows travel between ranks
    ('orders', 'cic_fcc_multigrid_pow2'),
    ('orders', 'cic_fcc_multigrid_vertex_pow2'),  # ... on vertex-centred grids (3 ghost layers)   # ... with interlacing + Fourier differentiation
    ('orders', 'tsc_bcc_n8_g16'), ('orders', 'tsc_bcc_deconv_down_n8_g16'), ('orders', 'pcs_fcc_fourier_n8_g16'),
    ('orders', 'ngp_fluid_n8_g16'),
    ('tiled_general', '-'),
    ('diff_orders', 'pm_n8_g16_d6'), ('diff_orders', 'pm_n8_g16_d1'),
    ('diff_orders', 'pm_edge_g16'), ('diff_orders', 'pm_n8_g16_vertex'),
    ('diff_orders', 'pm_n8_g16_deconv_up'), ('p3m_kick', 'p3m_n8_g32_plummer'),         # particles on cell / box / slab boundaries
    ('pp', 'pp_ewald_n4,pp'), ('pp', 'ppnonperiodic_n4,ppnonperiodic'),  # direct summation
    ('known', 'k2'), ('known', 'k3'),       # symmetric few-body configurations (pp, p3m)
    ('mixed_random', '-'),
    ('random_configs', '16'),               # the option space of gravity('pm'), drawn at random                  # particles + fluid (non-zero 𝒫) vs the oracle
    ('snapshot', '-'),                      # GADGET file -> Components over domains
    ('void', '-'),                          # ranks that start empty, first arrivals by exchange()
    ('k4', '16,2'), ('k4', '32,4'),
    # receivers not among the suppliers; slab faces of vertex-centred grids; populate() after
    # the rows have moved
    ('advice', 'cell'), ('advice', 'vertex'),
    # whole runs of the time loop (a_begin -> 1, ~140 base steps) against the reference's
    # (power-of-two meshes: the transposing FFT's sizes)
    ('traj', 'traj_pm_n8_g16'), ('traj', 'traj_p3m_n8_g32'),
    # the clustered box with five rungs populated: a slab holds most of a clump
    ('traj', 'traj_p3m_n16_g32_clustered'),
]


@pytest.mark.parametrize('world', [2, 4])
@pytest.mark.parametrize('case,arg', COMPONENT_CASES)
def test_components_over_domains(world, case, arg):
    _run_ranks(world, 'dist_component_worker.py', [case, arg])


def test_config3_shape_eight_slabs():
    """BASELINE configs[3]'s decomposition at an eighth of its linear size: 8 x-slab domains of
    64 layers of a 512^3 mesh, 256^3 particles, the streaming step (kick + drift + tile sort in
    one pass over particles in tile regions, leavers shipped by the pass: what bench.py runs
    on N > 1 GPUs), two steps, against one domain.  (The ranks share this GPU over gloo.)"""
    test_slab_domains_match_single_domain(8, 512, 'regions', n_side=256, steps=2)


def test_config4_shape_eight_slabs():
    """BASELINE configs[4]'s shape — particles by P³M on a 128^3 mesh, a fluid on a 64^3 grid,
    global PM grid 64: three mesh solves per long-range kick, the short-range kick with
    boundary suppliers, drift and exchange — on 8 domains against one (tests/test_gpu_fluid.py:
    test_config4_shape_across_domains)."""
    _run_ranks(8, 'dist_component_worker.py', ['config4', '48,64'], timeout=1500)


def test_rccl_one_rank_component_layer():
    """The production transport under the Component layer on the one GPU there is: a 1-rank
    RCCL group made the active decomposition (comm.init(force=True)), messages of the rank to
    itself sent through RCCL as well (CONCEPT_GPU_COMM_SELF=1).  Exercised calls: the particle
    exchange (all-to-all-v of rows with device-side destinations), the ghost-layer send /
    receive of the slab mesh, the transposes of the distributed solve, the P³M boundary-supplier
    shipping (sendrecv_component), the all-gathers of host(); results against the reference-
    generated goldens, as on gloo."""
    code = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path[:0] = [%r, %r]
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from concept_amd import comm
c = comm.init(force=True)
assert c is not None and not c.stage and c.self_transport and c.world == 1
def golden(name):
    return np.load(os.path.join(%r, 'tests', 'golden', name + '.npz'))
import test_gpu_p3m, test_gpu_pm, test_gpu_trajectory
test_gpu_pm.test_gravity_api_pm(None, golden)                       # PM kick, drift, exchange, sort
test_gpu_p3m.test_timeloop_sequence_vs_reference(golden, 'steps_p3m_n8_g32')  # + boundary suppliers
test_gpu_p3m.test_shortrange_two_components_receivers_not_suppliers(True)
test_gpu_trajectory.test_timeloop_run_vs_reference(golden, 'traj_pm_n8_g16')  # streaming + exchange
for seed in range(2):
    test_gpu_pm.test_random_streaming_timeloops(torch, seed)
torch.cuda.synchronize()
dist.destroy_process_group()
print('RCCL-COMPONENTS-OK')
''' % (REPO, os.path.join(REPO, 'tests'), REPO)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               CONCEPT_GPU_COMM_SELF='1')
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'RCCL-COMPONENTS-OK' in out, out[-4000:]


# Grids that are not a power of two on several domains (rocFFT per slab + a pack pass instead of
# the hand-written transposing passes; the reference takes any size divisible by the
# decomposition, communication.py:692-741, and its own tests use 24 and 36): the same bodies,
# against the same reference-generated goldens.  A slab must hold an even number (>= 4) of
# layers, which decides the admissible domain counts per case.
NON_POW2_CASES = [
    (2, 'p3m_kick', 'p3m_n12_g36_lattice'),      # 36: slabs of 18 layers
    (2, 'p3m_kick', 'p3m_n16_g48_clustered'), (4, 'p3m_kick', 'p3m_n16_g48_clustered'),
    (2, 'multigrid', 'multigrid_n8_g16'),        # 24 -> 16 -> 12, fluid on 8
    (2, 'multigrid', 'multigrid_n8_up32_down24'),
    (2, 'orders', 'cic_fcc_multigrid_n8'),
    (2, 'mixed', 'fluid2_pm_n6_g12'),
    (2, 'traj', 'traj_p3m_n8_g24'), (4, 'traj', 'traj_p3m_n8_g24'),   # whole runs, rungs
    (2, 'traj', 'traj_p3m_n8_g24_r1'),
]


@pytest.mark.parametrize('world,case,arg', NON_POW2_CASES)
def test_non_power_of_two_over_domains(world, case, arg):
    _run_ranks(world, 'dist_component_worker.py', [case, arg])


@pytest.mark.parametrize('world,N,mode', [(2, 48, False), (4, 96, 'regions'), (2, 72, 'fused'),
                                          (3, 36, False), (6, 96, 'regions')])
def test_slab_domains_match_single_domain_non_power_of_two(world, N, mode):
    """the low-level step (deposit, transposing solve, gather, drift, exchange, sort) on grids
    of 48, 72, 96 and 36 cells, also on 3 and 6 domains"""
    test_slab_domains_match_single_domain(world, N, mode)
