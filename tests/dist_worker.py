"""Worker of tests/test_gpu_distributed.py: one rank of a P-rank x-slab run.
All ranks share cuda:0 and talk over gloo (test only; production is one GPU
per rank over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    out_dir, N, n_side, steps, backend = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), \
        int(sys.argv[4]), sys.argv[5]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from concept_amd.distributed import (DistributedParticles, RegionParticles, SlabDomain,
                                         pm_kick, pm_step_regions, shortrange_kick)
    p3m = len(sys.argv) > 6 and sys.argv[6] == 'p3m'
    fused = len(sys.argv) > 6 and sys.argv[6] == 'fused'
    regions = len(sys.argv) > 6 and sys.argv[6] == 'regions'
    L = 64.0
    dom = SlabDomain(N, L)
    rng = np.random.default_rng(77)
    n = n_side**3
    pos = rng.uniform(0, L, (n, 3))
    cells_per_step = float(sys.argv[7]) if len(sys.argv) > 7 else 0.0
    mom = rng.normal(0, cells_per_step*(L/N)/0.9 if cells_per_step else 1.0, (n, 3))
    pos_d = torch.tensor(pos, device='cuda')
    owner = dom.mesh.owner_rank(pos_d).cpu().numpy()
    mine = np.nonzero(owner == rank)[0]
    parts = DistributedParticles(dom, pos_d[mine], torch.tensor(mom[mine], device='cuda'),
                                 torch.tensor(mine, device='cuda'))
    parts.tile_sort()
    contribution, C, kick, dtm = 0.37, -2.5, -0.002, 0.9
    if cells_per_step:
        contribution *= 1.4e-3   # (kicks about half the size of the thermal momenta)
    if regions:
        # the streaming form: kick + drift + tile sort in one pass, particles in tile regions
        # with gaps, leavers handed over by the pass itself
        rp = RegionParticles(parts)
        for step in range(steps):
            pm_step_regions(dom, rp, contribution, 4, C, kick, dtm, diff_order=2 + 2*(step % 2))
        rp.check()
        pos_o, mom_o, ids_o = rp.dense()
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), ids=ids_o.cpu().numpy(),
                 pos=pos_o.cpu().numpy(), mom=mom_o.cpu().numpy(), dens=np.zeros(1))
        dist.barrier()
        dist.destroy_process_group()
        return
    for step in range(steps):
        if p3m:
            scale = 1.25*L/N
            dm = shortrange_kick(dom, parts, scale=scale, range_=4.5*scale, tilesize=4.5*scale,
                                 tablesize=4096, softening=0.05*L/n_side, factor=3e-4)
            parts.view('mom').add_(dm)
        pm_kick(dom, parts, contribution, 4, C, kick, diff_order=2 + 2*(step % 2),
                long_range=p3m, E=-(2*np.pi/L*1.25*L/N)**2 if p3m else 0.0,
                next_dt_over_mass=dtm if fused and step % 3 != 2 else None)
        if fused:  # (every third step without a prepared histogram: both paths of the sort)
            parts.drift_exchange_sort(dtm)
        else:
            parts.drift(dtm)
            parts.exchange()
            parts.tile_sort()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), ids=parts.view('ids').cpu().numpy(),
             pos=parts.view('pos').cpu().numpy(), mom=parts.view('mom').cpu().numpy(),
             dens=np.zeros(1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
