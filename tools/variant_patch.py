"""A variant of libconcept_gpu.so with source text replaced, for A/B measurements that leave no
switch in the product:
    [VAR_SRC=cg_fft.hip] python tools/variant_patch.py NAME 'old text' 'new text' ['old2' 'new2' ...]
patches a copy of csrc/cg_shortrange.hip (or VAR_SRC), compiles it with the build's flags and
links it with the in-tree objects of the other sources into tools/_variants/NAME.so; load it with
CONCEPT_GPU_LIB=tools/_variants/NAME.so (concept_amd/lib.py)."""
import subprocess, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from concept_amd import build as b
b.build(verbose=False)
name, reps = sys.argv[1], sys.argv[2:]
fname = os.environ.get('VAR_SRC', 'cg_shortrange.hip')
src = open(b.CSRC + '/' + fname).read()
for o, n in zip(reps[::2], reps[1::2]):
    assert o in src, o
    src = src.replace(o, n)
os.makedirs(REPO + '/tools/_variants/src', exist_ok=True)
f = f'{REPO}/tools/_variants/src/{name}.hip'
open(f, 'w').write(src)
out = f'{REPO}/tools/_variants/{name}.so'; obj = out + '.o'
flags = list(b.FLAGS)
if fname == 'cg_fft.hip':
    flags[flags.index('-ffp-contract=off')] = '-ffp-contract=fast'
subprocess.check_call(['/opt/rocm/bin/hipcc'] + flags + ['-c', f, '-o', obj])
objs = [obj if s == fname else os.path.join(b.CSRC, s.replace('.hip', '.o')) for s in b.SOURCES]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + ['-L/opt/rocm/lib', '-lrocfft', '-Wl,-rpath,/opt/rocm/lib'])
print(out)
