"""Host resource checks (reference ``core/resources.py``): ``assignable`` (what
``Producer.to_array`` asks, :10-49) -- True when ``shape`` items of ``dtype`` fit in the
available memory minus a 50 MB margin, otherwise prints and returns False; ``allocate``
(:52-76) and ``pickleable`` (:79-98)."""

import numpy as np
import psutil


def assignable(shape, dtype=float, limit=None, msg=True):
    limit = psutil.virtual_memory().available if not limit else limit
    required = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if required < limit - int(50e6):
        return True
    if msg:
        name = getattr(dtype, "__name__", str(dtype))
        print(f"{tuple(shape)} type '{name}' requires {required / 1e9:.2f} GB "
              f"which exceeds the {limit / 1e9:.1f} GB available")
    return False


def allocate(jobs, requesting=None):
    """Physical cores to give ``jobs`` CPU-bound tasks (reference
    ``core/resources.py:52-76``): never more than asked for, than there are jobs, or than
    the process may run on once hyperthread siblings are counted as one."""
    wanted = jobs if requesting is None else requesting
    siblings = max(psutil.cpu_count() // max(psutil.cpu_count(logical=False) or 1, 1), 1)
    usable = len(psutil.Process().cpu_affinity()) // siblings
    return min(jobs, wanted, usable)


def pickleable(obj):
    """Does ``obj`` survive ``pickle.dumps``?  (reference ``core/resources.py:79-98``;
    what its concurrency tests ask of every producer and pipeline)"""
    import pickle
    try:
        pickle.dumps(obj)
    except Exception:      # noqa: BLE001 - any failure means "no"
        return False
    return True
