"""FIFOArray: the re-chunking queue behind GenProducer and MaskedProducer.

Semantics follow the reference's ``core/queues.py:9-70``: ``put`` appends
along the axis (the first put aliases the array), ``get`` pops ``chunksize``
samples, ``full`` means at least ``chunksize`` queued.  Instead of
re-concatenating the whole queue on every put (the reference's O(queue) copy,
queues.py:59-62), pieces are kept in a list; a pop copies exactly the chunk it
returns out of them (host arrays: a few threads) and leaves the remainder as a
view; the observable contents are identical.  Works for ndarrays and device
tensors.
"""

import numpy as np

from openseize_amd import _device as dev
from openseize_amd.core.arraytools import slice_along_axis, split_along_axis


class FIFOArray:
    def __init__(self, chunksize, axis):
        self._pieces = []
        self._size = 0
        self.chunksize = chunksize
        self.axis = axis

    # -- the reference exposes the joined queue as an attribute
    @property
    def queue(self):
        if not self._pieces:
            return np.array([])
        if len(self._pieces) > 1:
            self._pieces = [dev.concatenate(self._pieces, self.axis)]
        return self._pieces[0]

    @queue.setter
    def queue(self, value):
        if dev.size(value) == 0:
            self._pieces, self._size = [], 0
        else:
            self._pieces, self._size = [value], value.shape[self.axis]

    def qsize(self):
        return self._size

    def empty(self):
        return self._size == 0

    def full(self):
        return self._size >= self.chunksize

    def put(self, x):
        if dev.size(x) == 0:
            return
        self._pieces.append(x)
        self._size += x.shape[self.axis]

    def get(self):
        if len(self._pieces) <= 1:
            result, rest = split_along_axis(self.queue, self.chunksize, self.axis)
            self.queue = rest
            return result
        # several pieces queued: build the popped chunk alone (one copy of
        # chunksize samples, threaded for host arrays); what is left of the last
        # piece touched stays a view -- the queue itself is never re-joined
        need = min(self.chunksize, self._size)
        counts, left, keep = [], need, []
        for piece in self._pieces:
            n = piece.shape[self.axis]
            take = min(n, left)
            counts.append(take)
            left -= take
            if take < n:
                keep.append(piece if take == 0 else slice_along_axis(piece, take, None, axis=self.axis))
        result = dev.gather_along_axis(self._pieces, counts, self.axis)
        self._pieces, self._size = keep, self._size - need
        return result
