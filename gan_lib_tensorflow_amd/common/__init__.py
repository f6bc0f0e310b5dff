"""Counterpart of the reference's `common` package (imported there as `lib`), hot path only."""
from . import misc, ops  # noqa: F401
from ..store import ParamStore, get_default_store, set_default_store  # noqa: F401


def params_with_name(name):
    """common/__init__.py:40-41"""
    return get_default_store().params_with_name(name)
