"""The C ABI is the boundary, not PyTorch: INTEGRATION.md §1's reference-side stub run as
written — ctypes + hipMalloc/hipMemcpy from libamdhip64 — in an interpreter that imports neither
torch nor the concept_amd package."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_boundary_without_torch():
    p = subprocess.run([sys.executable, os.path.join(HERE, 'torchfree_boundary.py')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:'
                                + os.environ.get('LD_LIBRARY_PATH', '')))
    out = p.stdout.decode()
    assert p.returncode == 0 and 'TORCHFREE-OK' in out, out[-3000:]
