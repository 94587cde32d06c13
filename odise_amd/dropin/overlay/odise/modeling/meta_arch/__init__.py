"""Overlay shell of the reference's `odise.modeling.meta_arch` package (odise_amd.dropin): this directory first, the reference's own directory behind it."""
from odise_amd.dropin import chain_reference

__path__ = chain_reference(__name__, __path__)
from .odise import CaptionODISE, CategoryODISE  # noqa: E402,F401  (the reference's package exports, odise/modeling/meta_arch/__init__.py)

__all__ = ["CategoryODISE", "CaptionODISE"]
