"""Build libconcept_gpu.so (gfx950) in-tree with hipcc.

`python -m concept_amd.build` or `__graft_entry__.build()`.  The .so is
git-ignored but travels to the GPU box with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libconcept_gpu.so')
SOURCES = ['cg_context.hip', 'cg_mesh_kernels.hip', 'cg_tiled_kernels.hip', 'cg_fft.hip', 'cg_shortrange.hip', 'cg_shortrange_dense.hip', 'cg_rungs.hip',
           'cg_particles.hip', 'cg_general.hip', 'cg_pp.hip']
HEADERS = [os.path.join(CSRC, 'cg_internal.h'), os.path.join(CSRC, 'cg_kspace.h'), os.path.join(CSRC, 'cg_tiles.h'), os.path.join(CSRC, 'cg_substep.h'), os.path.join(REPO, 'include', 'concept_gpu.h')]

FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
    # the reference's operation order is the parity contract: no FMA contraction
    '-ffp-contract=off',
    # hardware FP64 atomic add (global_atomic_add_f64 / ds_add_f64) instead of CAS loops
    '-munsafe-fp-atomics',
    '-Wall', '-Wno-unused-result',
    '-I' + os.path.join(REPO, 'include'), '-I' + CSRC, '-I/opt/rocm/include',
]


def _code_only(text):
    """the source text without comments and with white space collapsed: what the compiler sees"""
    import re
    pattern = r'("(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\')|//[^\n]*|/\*.*?\*/'
    text = re.sub(pattern, lambda m: m.group(1) or ' ', text, flags=re.S)
    return ' '.join(text.split())


def source_hash():
    """sha256 over the CODE libconcept_gpu.so is built from (csrc/*.hip, *.h and the C ABI header
    without their comments, and the compiler flags): written beside the library at build time
    (libconcept_gpu.so.srchash) and stamped into profiles/*_pmc_hbm_traffic.json by
    tools/pmc_traffic.py, so that bench.py can tell whether a committed counter run describes the
    kernels it is timing (a reworded comment does not make it a different library)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                   if f.endswith(('.hip', '.h'))) + [HEADERS[-1]]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'r', encoding='utf-8') as fh:
            h.update(_code_only(fh.read()).encode())
    h.update(' '.join(FLAGS[:7]).encode())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + HEADERS + [os.path.abspath(__file__)]):
            flags = list(FLAGS)
            if src == 'cg_fft.hip':
                # the FFT butterflies have no reference operation order to keep (the
                # reference calls FFTW / pocketfft): let them contract to FMA
                flags[flags.index('-ffp-contract=off')] = '-ffp-contract=fast'
            cmd = [hipcc] + flags + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))   # (the sources compile side by side)
        objs.append(o)
    for cmd, job in jobs:
        if job.wait():
            raise subprocess.CalledProcessError(job.returncode, cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [
            '-L/opt/rocm/lib', '-lrocfft', '-Wl,-rpath,/opt/rocm/lib']
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(LIB + '.srchash', 'w') as f:
            f.write(source_hash() + '\n')
    if not os.path.exists(LIB + '.srchash'):
        with open(LIB + '.srchash', 'w') as f:
            f.write(source_hash() + '\n')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
