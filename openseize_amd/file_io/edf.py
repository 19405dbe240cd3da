"""EDF reader whose record decode runs on the device.

Host side: header parsing and record location, same interface and semantics
as the reference's ``file_io/edf.py`` (``Header`` :111-314, ``Reader``
:317-586): ``Reader(path)``, ``.header``, ``.channels`` (settable), ``.shape``,
``.read(start, stop=None, padvalue=nan)``, ``open()`` / ``close()`` and context
management.  Device side: the little-endian int16 records are uploaded as they
are (2 B per sample over PCIe instead of 8) and de-interleaved + scaled by
``osz_edf_decode``.  ``read(..., device=True)`` returns a CUDA tensor so that a
``producer(reader, chunksize, axis=-1, device=True)`` chain never holds float64
samples on the host.  Writing EDF files and annotations are out of scope.
"""

import copy
import ctypes
from pathlib import Path

import numpy as np

from openseize_amd import _device as dev
from openseize_amd import _lib


class Header(dict):
    """Dictionary of the EDF header fields with '.' access (reference
    file_io/bases.py:26-120, file_io/edf.py:111-314)."""

    def __init__(self, path):
        self.path = Path(path) if path else None
        dict.__init__(self)
        self.update(self.read())

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(
                f"'{type(self).__name__}' object has no attribute '{name}'") from exc

    def bytemap(self, num_signals=None):
        """Field -> ([byte counts], type) of the EDF specification
        (edf.py:123-162)."""
        ns = self.count_signals() if num_signals is None else num_signals
        return {
            "version": ([8], str), "patient": ([80], str), "recording": ([80], str),
            "start_date": ([8], str), "start_time": ([8], str),
            "header_bytes": ([8], int), "reserved_0": ([44], str),
            "num_records": ([8], int), "record_duration": ([8], float),
            "num_signals": ([4], int),
            "names": ([16] * ns, str), "transducers": ([80] * ns, str),
            "physical_dim": ([8] * ns, str),
            "physical_min": ([8] * ns, float), "physical_max": ([8] * ns, float),
            "digital_min": ([8] * ns, float), "digital_max": ([8] * ns, float),
            "prefiltering": ([80] * ns, str),
            "samples_per_record": ([8] * ns, int), "reserved_1": ([32] * ns, str),
        }

    def count_signals(self):
        if not self.path:
            return int(self["num_signals"])
        with open(self.path, "rb") as fp:
            fp.seek(252)
            return int(fp.read(4).strip().decode())

    def read(self, encoding="ascii"):
        header = {}
        if not self.path:
            return header
        with open(self.path, "rb") as fp:
            for name, (nbytes, dtype) in self.bytemap().items():
                res = [dtype(fp.read(n).strip().decode(encoding=encoding))
                       for n in nbytes]
                # per-signal fields stay lists even for a single signal
                header[name] = res[0] if len(nbytes) == 1 and name not in _PER_SIGNAL else res
        return header

    # -- derived quantities (edf.py:200-300)
    @property
    def annotated(self):
        return "EDF Annotations" in self.names

    @property
    def annotation(self):
        return self.names.index("EDF Annotations") if self.annotated else None

    @property
    def channels(self):
        signals = list(range(self.num_signals))
        if self.annotation:
            signals.pop(self.annotation)
        return signals

    @property
    def samples(self):
        samples = np.array(self.samples_per_record) * self.num_records
        return [samples[ch] for ch in self.channels]

    @property
    def record_map(self):
        cum = np.cumsum(np.insert(self.samples_per_record, 0, 0))
        return [slice(a, b) for a, b in zip(cum, cum[1:])]

    @property
    def slopes(self):
        ch = self.channels
        pmax, pmin = np.array(self.physical_max)[ch], np.array(self.physical_min)[ch]
        dmax, dmin = np.array(self.digital_max)[ch], np.array(self.digital_min)[ch]
        return (pmax - pmin) / (dmax - dmin)

    @property
    def offsets(self):
        ch = self.channels
        pmin, dmin = np.array(self.physical_min)[ch], np.array(self.digital_min)[ch]
        return pmin - self.slopes * dmin

    def filter(self, indices):
        header = copy.deepcopy(self)
        for key, value in header.items():
            if isinstance(value, list):
                header[key] = [value[idx] for idx in indices]
        bytemap = self.bytemap(len(indices))
        header["header_bytes"] = sum(sum(tup[0]) for tup in bytemap.values())
        header["num_signals"] = len(indices)
        return header


_PER_SIGNAL = {"names", "transducers", "physical_dim", "physical_min", "physical_max",
               "digital_min", "digital_max", "prefiltering", "samples_per_record",
               "reserved_1"}


class Reader:
    """Reader of EDF / EDF+ data records (reference edf.py:317-586) with the
    decode on the device."""

    def __init__(self, path):
        self.path = Path(path)
        self.mode = "rb"
        self._fobj = open(self.path, self.mode)
        self.header = Header(path)
        self._channels = self.header.channels

    # -- file handle (file_io/bases.py Reader)
    def open(self):
        if self._fobj is None or self._fobj.closed:
            self._fobj = open(self.path, self.mode)

    def close(self):
        if self._fobj and not self._fobj.closed:
            self._fobj.close()

    def __enter__(self):
        return self

    # a reader travels between processes closed; it reopens on first use
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_fobj"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __exit__(self, exc_type, exc_value, traceback):
        self.close()

    @property
    def channels(self):
        return self._channels

    @channels.setter
    def channels(self, values):
        if not isinstance(values, (list, tuple, range)):
            raise ValueError("Channels must be type Sequence not {}".format(type(values)))
        self._channels = values

    @property
    def shape(self):
        return len(self.channels), max(self.header.samples)

    # -- record location (edf.py:421-483), pure host logic
    def plan(self, start, stop, channels):
        """Everything ``read`` needs besides the bytes: the union record range
        to load and, per channel, where its samples sit and how many it can
        deliver (a channel with a lower sample rate runs out first)."""
        hdr = self.header
        spr_all = np.array(hdr.samples_per_record)
        spr = spr_all[list(channels)]
        nrec = hdr.num_records
        r0 = start // spr
        r1 = np.minimum(np.ceil(stop / spr).astype(int), nrec)
        a = start - r0 * spr
        avail = np.maximum(r1 - r0, 0) * spr
        lens = np.maximum(np.minimum(a + (stop - start), avail) - a, 0)
        rec0 = int(min(r0.min(), nrec))
        rec1 = int(max(r1.max(), rec0))
        choff = np.cumsum(np.insert(spr_all, 0, 0))[list(channels)]
        return {"rec0": rec0, "nrec": rec1 - rec0, "reclen": int(spr_all.sum()),
                "spr": spr.astype(np.int32), "choff": choff.astype(np.int32),
                "len": lens.astype(np.int64), "width": int(lens.max()) if len(lens) else 0}

    def _records(self, a, cnt):
        """Raw int16 records [a, a + cnt) exactly as stored (edf.py:452-483)."""
        hdr = self.header
        reclen = sum(hdr.samples_per_record)
        offset = hdr.header_bytes + a * reclen * 2
        self._fobj.seek(0)
        return np.fromfile(self._fobj, "<i2", cnt * reclen, offset=offset)

    def read(self, start, stop=None, padvalue=np.nan, device=False):
        """Samples [start, stop) of this reader's channels as a float64
        (channels, samples) array (edf.py:558-586); ``device=True`` returns a
        CUDA tensor instead of an ndarray."""
        import torch
        nchan = len(self.channels)
        if start > max(self.header.samples):
            empty = np.empty((nchan, 0))
            return torch.from_numpy(empty).cuda() if device else empty
        if not stop:
            stop = max(self.header.samples)
        start, stop = int(start), int(stop)
        self.open()
        p = self.plan(start, stop, self.channels)
        lib = dev.require_gpu()
        idx = [self.header.channels.index(c) for c in self.channels]
        raw = torch.from_numpy(self._records(p["rec0"], p["nrec"])).cuda()
        if raw.numel() == 0:
            raw = torch.zeros(1, dtype=torch.int16, device="cuda")
        t = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        choff, spr, lens = t(p["choff"]), t(p["spr"]), t(p["len"])
        slope, offset = t(self.header.slopes[idx]), t(self.header.offsets[idx])
        out = torch.empty((nchan, p["width"]), dtype=torch.float64, device="cuda")
        _lib.check(lib.osz_edf_decode(
            dev.ptr(raw), p["reclen"], nchan, dev.ptr(choff), dev.ptr(spr),
            dev.ptr(slope), dev.ptr(offset), dev.ptr(lens), p["rec0"], start,
            p["width"], ctypes.c_double(padvalue), dev.ptr(out),
            max(out.stride(0), 1), dev.stream_ptr()))
        return out if device else out.cpu().numpy()
