"""PMC probe of the fused drift + sort scatter: HBM write traffic with and without the id
stream, for particles that barely move (the bench situation)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd.mesh import PotentialMesh
mode = sys.argv[1] if len(sys.argv) > 1 else 'ids'
n, g = 2**26, 512
box = 512.0
m = PotentialMesh(g, box)
torch.manual_seed(1)
pos = torch.rand((n, 3), dtype=torch.float64, device='cuda')*box
mom = torch.randn((n, 3), dtype=torch.float64, device='cuda')*1e-3
ids = torch.arange(n, dtype=torch.int64, device='cuda')
po, mo, io = torch.empty_like(pos), torch.empty_like(mom), torch.empty_like(ids)
tab = m.new_tile_table()
m.sort_particles(pos, mom, ids, po, mo, io, tab)
for _ in range(3):
    if mode == 'ids':
        m.drift_sort(po, mo, io, pos, mom, ids, 1e-3, tab)
        m.drift_sort(pos, mom, ids, po, mo, io, 1e-3, tab)
    else:
        m.drift_sort(po, mo, None, pos, mom, None, 1e-3, tab)
        m.drift_sort(pos, mom, None, po, mo, None, 1e-3, tab)
torch.cuda.synchronize()
print('done', mode)
