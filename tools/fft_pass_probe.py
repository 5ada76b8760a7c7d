"""Per-pass timing of the hand-written FFT at 1024^3: forward (z, y, x), backward (x, y, z)
and the fused solve, to separate the cost of the x stride from the fused pass's compute."""
import sys, os, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from concept_amd.mesh import PotentialMesh
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = PotentialMesh(g, 1000.0)
m.zero()
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
out = {'forward_zyx': t(lambda: m.poisson_forward(0, 0.0, apply_kernel=False)),
       'backward_xyz': t(lambda: m.poisson_backward()),
       'kernel_only': t(lambda: m.poisson_kernel(4, -1.0)),
       'solve_passes': m.poisson_solve_timed(4, -1.0)}
print(json.dumps(out))
