"""meterelf_amd -- MI355X-native drop-in for meterelf's per-image hot path.

Public surface = the reference's (meterelf/__init__.py:1-6) plus the batched
engine: `MeterReader` (one GPU) and `meterelf_amd._dist` (one process per GPU).
"""
from ._api import MeterImageData, get_meter_values, release_cached_contexts
from ._engine import MeterReader

__all__ = ['MeterImageData', 'get_meter_values', 'MeterReader', 'release_cached_contexts']
