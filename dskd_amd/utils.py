"""Small host-side helpers."""
from collections import OrderedDict

import torch

_CONST_CACHE = OrderedDict()
_CONST_CACHE_MAX = 512


def device_const(values, dtype, device):
    """A small constant tensor built from python numbers, cached per (values, dtype, device).
    Repeated steps reuse the device copy instead of issuing a host->device transfer (which is
    also what makes a step capturable into a hipGraph: no H2D inside the captured region).
    Never modify the returned tensor in place."""
    def freeze(v):
        return tuple(freeze(x) for x in v) if isinstance(v, (list, tuple)) else v
    key = (freeze(values), dtype, str(device))
    t = _CONST_CACHE.get(key)
    if t is None:
        t = torch.tensor(values, dtype=dtype, device=device)
        _CONST_CACHE[key] = t
        if len(_CONST_CACHE) > _CONST_CACHE_MAX:
            _CONST_CACHE.popitem(last=False)
    else:
        _CONST_CACHE.move_to_end(key)
    return t


def const_cache_snapshot():
    """References to every cached constant: a captured hipGraph holds raw pointers to them, so
    the graph's owner keeps this list alive (the LRU may otherwise evict and free them)."""
    return list(_CONST_CACHE.values())
