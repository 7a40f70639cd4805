"""A small LRU for the CPU oracle's results inside one pytest process.

The GPU parity files run the same (shape, seed, mode) case under several forced kernel strategies; the oracle's fp64 /
fp32 results do not depend on the strategy, and the oracle — tens of eager CPU ops, twice — is most of a case's time.
tests/conftest.py orders the items so that the strategies of one case run back to back; a handful of entries is enough.
Nothing about WHAT is compared changes: every test still checks the device results against the oracle's."""
import collections

_LRU = collections.OrderedDict()
_MAX = 8
hits = misses = 0


def memo(key, fn):
    global hits, misses
    if key in _LRU:
        _LRU.move_to_end(key)
        hits += 1
        return _LRU[key]
    misses += 1
    val = fn()
    _LRU[key] = val
    while len(_LRU) > _MAX:
        _LRU.popitem(last=False)
    return val
