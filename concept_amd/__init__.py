"""concept_amd — MI355X-native PM/P3M gravity stepper behind CO*N*CEPT's
gravity(method, receivers, suppliers, ᔑdt, interaction_type, printout) API.

Every module of the path (mesh, species, interactions) imports
concept_amd.lib, which loads libconcept_gpu.so and raises if it has not been
built (`python -m concept_amd.build`): there is no CPU fallback.  Only
`concept_amd.build` and `concept_amd.commons` are importable without it."""
from . import commons  # noqa: F401
