#!/usr/bin/env python3
"""Run the repository's randomised differential tests beyond their committed seeds:
    python tools/fuzz_gpu.py FIRST LAST        (seeds FIRST .. LAST-1 of each)
 * tests/test_gpu_pm.py    test_random_streaming_timeloops      (streaming loop vs separate passes)
 * tests/test_gpu_p3m.py   test_random_shortrange_vs_oracle     (short-range sweep vs CPU oracle)
 * tests/test_gpu_fluid.py test_random_configurations_vs_oracle (option space of gravity('pm'))
Prints one line per failure with the seed; exit status 1 if any."""
import os
import sys
import traceback
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', 'tests'))
import torch  # noqa: E402
import test_gpu_fluid  # noqa: E402
import test_gpu_p3m  # noqa: E402
import test_gpu_pm  # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
cases = [('streaming', lambda s: test_gpu_pm.test_random_streaming_timeloops(torch, s)),
         ('shortrange', test_gpu_p3m.test_random_shortrange_vs_oracle),
         ('configs', test_gpu_fluid.test_random_configurations_vs_oracle)]
only = set(sys.argv[3].split(',')) if len(sys.argv) > 3 else None
bad = 0
for name, fn in cases:
    if only and name not in only:
        continue
    for seed in range(first, last):
        try:
            fn(seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f'FAIL {name} seed {seed}: {type(e).__name__}: {str(e)[:300]}', flush=True)
            traceback.print_exc(limit=3)
    print(f'{name}: seeds {first}..{last - 1} done', flush=True)
sys.exit(1 if bad else 0)
