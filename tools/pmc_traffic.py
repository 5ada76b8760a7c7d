#!/usr/bin/env python3
"""profiles/rNN_pmc_hbm_traffic.json from two rocprofv3 --pmc runs of the bench command
(FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes, as MI355X_MICROARCH.md prescribes):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d DIR_F -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d DIR_W -- python bench.py ...
    python tools/pmc_traffic.py DIR_F DIR_W OUT.json WORKLOAD_NAME COMMIT "COMMAND"

Units and corrections (guide, HBM section): the counters are KiB of L2 <-> fabric requests
(Infinity-Cache hits included); on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so
reads are doubled (calibrated in round 1 on k_drift: 12.88 GB read showed as 6.44); WRITE_SIZE
is taken as is.  Per launch = the minimum over the calls of the run (the first call of a kernel
writing never-touched pages counts the page first-touch)."""
import collections
import datetime
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BENCH_KEYS = {  # bench.py kernel names -> demangled prefixes of the device kernels
    'kick_drift_sort': ['k_gather_kick_tiled<2, 16, 2, false>'],
    'fft_zy_forward_chunked': [], 'fft_yz_backward_chunked': [],
    'gather_kick': ['k_gather_kick_tiled<2, 16, 1, false>', 'k_gather_kick_tiled<2, 16, 0, false>'],
    'deposit': ['k_deposit_cic_pull<16, false>'],
    'fft_x_fused_kspace': ['k_fft_strided_h<10, 256, 2,', 'k_fft_strided_p<10, 512, 2, 8>'],
    'sr_sweep': ['k_sr_sweep_blocks'],
}


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for db in glob.glob(os.path.join(path, '**', '*.db'), recursive=True):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute('pragma table_info(counters_collection)')]
        ci = {c: i for i, c in enumerate(cols)}
        namec = 'kernel_name' if 'kernel_name' in ci else 'name'
        for r in con.execute('select * from counters_collection'):
            if r[ci['counter_name']] == counter:
                acc[r[ci[namec]]].append(float(r[ci['value']]))
    return acc


def main():
    dir_f, dir_w, out, workload, commit, command = sys.argv[1:7]
    f, w = per_kernel(dir_f, 'FETCH_SIZE'), per_kernel(dir_w, 'WRITE_SIZE')
    kernels = {}
    for name in sorted(set(f) | set(w), key=lambda k: -sum(f.get(k, [0]))):
        short = name.split('(')[0].replace('void ', '')
        rd, wr = f.get(name, [0.0]), w.get(name, [0.0])
        e = {'calls': max(len(rd), len(wr)),
             'read_GB': round(min(rd)*1024*2/1e9, 3), 'write_GB': round(min(wr)*1024/1e9, 3),
             'read_GB_mean': round(sum(rd)/len(rd)*1024*2/1e9, 3),
             'write_GB_mean': round(sum(wr)/len(wr)*1024/1e9, 3), 'fetch_corr': 2.0}
        e['total_GB'] = round(e['read_GB'] + e['write_GB'], 3)
        for key, prefixes in BENCH_KEYS.items():
            if any(short.startswith(p) for p in prefixes):
                e['bench_key'] = key
        kernels[short] = e
    from concept_amd import build as cg_build
    json.dump({'workload_name': workload, 'commit': commit,
               # what the library that ran was built from (bench.py quotes these figures only
               # for a library with the same hash)
               'csrc_hash': open(cg_build.LIB + '.srchash').read().strip(),
               'date': datetime.date.today().isoformat(), 'command': command,
               'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate '
                         'passes); KiB of L2 <-> fabric requests, reads doubled (gfx950), per '
                         'launch = minimum over the calls', 'kernels': kernels},
              open(out, 'w'), indent=1)
    for k, e in list(kernels.items())[:12]:
        print(f"{k[:70]:70s} calls {e['calls']:4d} read {e['read_GB']:8.3f} write "
              f"{e['write_GB']:8.3f} GB {e.get('bench_key', '')}")


if __name__ == '__main__':
    main()
