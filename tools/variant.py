#!/usr/bin/env python3
"""Build a variant of libconcept_gpu.so for kernel A/B experiments:
    python tools/variant.py OUT.so cg_shortrange.hip -DCG_SR_BATCH8 [more flags]
recompiles that one source with the extra flags and links it with the in-tree objects of the
others.  Load it with CONCEPT_GPU_LIB=OUT.so (concept_amd/lib.py)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from concept_amd import build as b  # noqa: E402

out, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build(verbose=False)
flags = list(b.FLAGS)
if src == 'cg_fft.hip':
    flags[flags.index('-ffp-contract=off')] = '-ffp-contract=fast'
obj = out + '.' + src.replace('.hip', '.o')
subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + flags + extra +
                      ['-c', os.path.join(b.CSRC, src), '-o', obj])
objs = [obj if s == src else os.path.join(b.CSRC, s.replace('.hip', '.o')) for s in b.SOURCES]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out]
                      + objs + ['-L/opt/rocm/lib', '-lrocfft', '-Wl,-rpath,/opt/rocm/lib'])
print(out)
