"""concept_amd — MI355X-native PM/P3M gravity stepper behind CO*N*CEPT's
gravity(method, receivers, suppliers, ᔑdt, interaction_type, printout) API.

Importing the package loads libconcept_gpu.so (concept_amd/lib.py) and fails
loudly if it has not been built: there is no CPU fallback."""
from . import commons  # noqa: F401
from . import lib  # noqa: F401  (raises if the HIP library is missing)
