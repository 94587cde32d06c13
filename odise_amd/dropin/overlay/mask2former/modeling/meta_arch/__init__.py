"""Overlay shell of the reference's `mask2former.modeling.meta_arch` package (odise_amd.dropin): this directory first, the reference's own directory behind it."""
from odise_amd.dropin import chain_reference

__path__ = chain_reference(__name__, __path__)
