#!/usr/bin/env python3
"""Time the transposing (x-slab) Poisson solve with ONE rank under RCCL (CONCEPT_GPU_DIST_FORCE):
the FFT passes of the multi-GPU path — y passes writing / reading the all-to-all buffers blocked
by destination domain — without the links.  usage: dist_fft_probe.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ['CONCEPT_GPU_DIST_FORCE'] = '1'
import torch
import torch.distributed as dist
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from concept_amd.distributed import SlabDomain
dom = SlabDomain(N, float(N))
dom.mesh.zero()
ms = []
for i in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dom.poisson_solve(4, -2.5)
    e1.record()
    torch.cuda.synchronize()
    ms.append(round(e0.elapsed_time(e1), 3))
sys.stderr.flush()
print('N', N, 'pieces', len(dom.pieces), 'transposing solve ms:', ms,
      'split', os.environ.get('CONCEPT_GPU_FFT_SPLIT', 'default'), file=sys.stderr)
dist.destroy_process_group()
