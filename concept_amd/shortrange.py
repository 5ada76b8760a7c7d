"""P3M short-range tile sweep (host side) — to be built on
libconcept_gpu.so's cg_shortrange_* entry points."""
from .lib import ConceptGPUError


def component_component(force, receivers, suppliers, ᔑdt_rungs, gridsize):
    raise ConceptGPUError('the P3M short-range tile sweep is not built yet')
