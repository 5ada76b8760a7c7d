"""Time cg_poisson_solve by itself at 1024^3 (HIP events around 10 solves), for A/B runs of
variant builds (tools/variant_patch.py with VAR_SRC=cg_fft.hip, loaded through CONCEPT_GPU_LIB)
or of CONCEPT_GPU_FFT / CONCEPT_GPU_FFT_SPLIT."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd.mesh import PotentialMesh
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = PotentialMesh(g, float(g))
m.zero()
f = lambda: m.poisson_solve(4, -1.0, False, 0.0)
for _ in range(3): f()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1)/10)
print(' '.join(f'{k}={v}' for k, v in os.environ.items() if k.startswith('CONCEPT_GPU_FFT')), 'solve ms:', ' '.join(f'{t:.3f}' for t in ts))
