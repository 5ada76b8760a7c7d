"""Lane schedule of the short-range sweep's pair loop on a uniform box, simulated (numpy): the
sweep as built — a receiver group is a cell column of a tile (2 cells along z), its suppliers the
5 x 5 columns x 6 cells around it — against groups of ONE cell with the cell beyond reach left
out of every column (5 x 5 x 5 cells: what a skip of whole trips by the receivers' z extent can
reach at best).  Counts 64-lane trips of the loop as sr_cell_ranges walks them
(2 S suppliers per full trip, ranges joined across their ends), executed tests, and the groups'
fixed work.  `python tools/sr_ztrip_sim.py [particles per tile]`"""
import sys
import numpy as np

rng = np.random.default_rng(1)
per_tile = float(sys.argv[1]) if len(sys.argv) > 1 else 22.26
lam = per_tile/8            # per half-tile cell
G = 40000                   # groups simulated


def trips(n_supp, R):
    S = np.minimum(64//R, 32)
    return np.ceil(n_supp/(2*S)), S


def scheme(cells_recv, cells_per_column, label):
    R = rng.poisson(lam*cells_recv, G)
    R = R[R > 0]
    n = rng.poisson(lam*cells_per_column*25, R.size)      # suppliers of the 5 x 5 columns
    t, S = trips(n, np.minimum(R, 64))
    lanes = 2*64*t          # (two pairs per lane and trip)
    tests = n*R
    hits = tests*(4/3*np.pi*1.0**3)/((5*0.5)**2*cells_per_column*0.5)  # sphere of the range over the window (range = tile = 2 cells)
    per_recv = t.sum()/R.sum()
    print(f'{label}: {R.mean():.2f} receivers per group, {n.mean():.0f} suppliers, '
          f'{per_recv:.2f} trips per receiver, tests per receiver {tests.sum()/R.sum():.0f}, '
          f'lane use {tests.sum()/ (lanes*1.0).sum():.3f}, groups per receiver {R.size/R.sum():.3f}')
    return per_recv, R.size/R.sum()


a = scheme(2, 6, 'as built (2-cell groups, 6 cells per column)')
b = scheme(1, 5, 'one-cell groups, 5 cells per column     ')
print(f'trips per receiver {b[0]/a[0]:.3f} x, groups (chunk loads, folds, Δmom read-modify-writes, '
      f'range bounds: 25 segments instead of 5) {b[1]/a[1]:.2f} x')
