"""Thread-count sweep of the all-cores CPU baseline (oracle OpenMP build) on the bench box."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
oracle.build()
L_ = oracle.lib('omp')
out = {}
for sample_n, grid, steps in ((128, 256, 3), (256, 512, 1)):
    n = sample_n**3
    L = float(grid)
    for threads in (8, 16, 32, 64, 128):
        L_.orc_threads(threads)
        rng = np.random.default_rng(7)
        pos = rng.uniform(0, L, (n, 3)); mom = np.zeros((n, 3))
        def run(k):
            t0 = time.perf_counter()
            for _ in range(k):
                oracle.drift(pos, mom, 1e-3, L, fast='omp')
                oracle.pm_long_range(pos, mom, mass=1.0, boxsize=L, gridsize=grid, G_Newton=1.0,
                                     dt_1=1e-3, dt_dens=1e-3, dt_kick=1e-3, diff_order=2,
                                     fast='omp', want_indices=False)
            return time.perf_counter() - t0
        run(1)
        dt = run(steps)
        out[f'{sample_n}/{grid}/t{threads}'] = round(n*steps/dt/1e6, 3)
        print(sample_n, grid, threads, out[f'{sample_n}/{grid}/t{threads}'], flush=True)
print(json.dumps(out))
