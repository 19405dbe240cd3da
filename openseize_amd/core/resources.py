"""Host resource check used by ``Producer.to_array`` (reference
``core/resources.py:10-49``): True when ``shape`` items of ``dtype`` fit in
the available memory minus a 50 MB margin; otherwise prints and returns False."""

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
