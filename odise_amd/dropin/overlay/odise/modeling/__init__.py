"""Overlay shell of the reference's `odise.modeling` package (odise_amd.dropin): this directory first, the reference's own directory behind it."""
from odise_amd.dropin import chain_reference

__path__ = chain_reference(__name__, __path__)
