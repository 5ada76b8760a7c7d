"""Import the reference's pure-Python hot-path modules in THIS container.

Test infrastructure only.  Used by tests/golden/make_golden.py to produce the
committed golden vectors; it reads /root/reference (which does not exist on
the GPU box) and therefore is never imported by tests, bench.py, smoke() or
the product package.

Nothing in the reference is edited.  Four driver-side shims:
  1. a one-rank `mpi4py` stand-in (oracle/refharness/mpi4py),
  2. an identity `blessings` stand-in (oracle/refharness/blessings.py),
  3. `np.compat.py3k` pre-created for NumPy >= 2 (commons.py:486-488),
  4. `warnings.catch_warnings(action=..., category=...)` accepted on
     Python 3.10 (commons.py:568,618,1813 use the 3.11 signature).
Parameters are injected the reference's own way: a `.path` file in the
working directory (commons.py:1700-1721) and the param text at
<job_dir>/<jobid>/param (commons.py:1754-1784).
"""
import os
import sys
import types
import warnings

REFERENCE = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))

_PATH_KEYS_DIRS = [
    'concept_dir', 'build_dir', 'dep_dir', 'doc_dir', 'ic_dir', 'job_dir', 'output_dir',
    'param_dir', 'reusable_dir', 'src_dir', 'test_dir', 'tmp_dir', 'util_dir',
]


def _write_path_file(workdir):
    """Rewrite the reference's .path with every path moved under workdir
    (src_dir keeps pointing at the read-only reference sources)."""
    lines = []
    with open(f'{REFERENCE}/.path', encoding='utf-8') as f:
        for line in f:
            s = line.strip()
            if not s or s.startswith('#') or '=' not in s:
                continue
            key, val = s.split('=', 1)
            val = val.strip().strip('\'"')
            tail = val.split('/concept', 1)[1] if '/concept' in val else '/' + key
            if key == 'src_dir':
                new = f'{REFERENCE}/src'
            elif key == 'concept_dir':
                new = workdir
            else:
                new = workdir + tail
            lines.append(f"{key}='{new}'")
    with open(f'{workdir}/.path', 'w', encoding='utf-8') as f:
        f.write('\n'.join(lines) + '\n')
    for key in ('job', 'output', 'ic', '.reusable', '.tmp', 'param'):
        os.makedirs(f'{workdir}/{key}', exist_ok=True)


def load_reference(param_text, workdir):
    """Returns a namespace of the imported reference modules.  One call per
    process (the reference keeps its parameters as module globals)."""
    os.makedirs(workdir, exist_ok=True)
    _write_path_file(workdir)
    os.makedirs(f'{workdir}/job/1', exist_ok=True)
    with open(f'{workdir}/job/1/param', 'w', encoding='utf-8') as f:
        f.write(param_text)
    os.chdir(workdir)
    sys.dont_write_bytecode = True
    sys.argv = ['ref', 'jobid=1', "param='golden'"]
    sys.path.insert(0, f'{REFERENCE}/src')
    sys.path.insert(0, HERE)
    # shim 3
    import numpy as np
    if not hasattr(np, 'compat'):
        np.compat = types.SimpleNamespace()
    try:
        np.compat.py3k
    except Exception:
        np.compat = types.SimpleNamespace(py3k=types.SimpleNamespace())
    # shim 4
    if sys.version_info < (3, 11):
        _orig = warnings.catch_warnings

        class _CatchWarnings(_orig):
            def __init__(self, *, record=False, module=None, action=None, category=Warning,
                         lineno=0, append=False):
                super().__init__(record=record, module=module)
                self._gx_action = action
                self._gx_category = category

            def __enter__(self):
                r = super().__enter__()
                if self._gx_action is not None:
                    warnings.simplefilter(self._gx_action, self._gx_category)
                return r

        warnings.catch_warnings = _CatchWarnings
    import importlib
    mods = {}
    for name in ('commons', 'communication', 'mesh', 'integration', 'species', 'ewald',
                 'interactions', 'gravity'):
        mods[name] = importlib.import_module(name)
    return types.SimpleNamespace(**mods)
