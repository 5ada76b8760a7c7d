"""Single-rank stand-in for mpi4py, used ONLY by the golden-vector generator
(tests/golden/make_golden.py) to import the pure-Python reference in this
container.  It is test infrastructure: nothing in concept_amd/ imports it and
it never travels as part of the product path."""
from . import rc  # noqa: F401
