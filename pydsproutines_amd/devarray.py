"""Minimal device-array type for the cupy-signature entry points.

The reference's GPU functions take and return ``cupy.ndarray``
(xcorrRoutines.py:169-371, cupyExtensions.py) and its unit test requires a
TypeError for host arrays (xcorrRoutines.py:2150-2157).  cupy does not exist on
ROCm here and must not be shimmed, so the host layer ships this small HBM-backed
array: contiguous, C-order, allocated with ``caf_malloc`` (a caching allocator over
hipMalloc, the counterpart of cupy's default memory pool), with the
handful of members the reference code paths touch: ``shape / dtype / size /
ndim / nbytes / get() / reshape() / [row or range slicing] / conj() / copy()``.

``as_device_ptr`` also accepts torch CUDA tensors (``data_ptr()``), which is how
bench.py hands over HBM-resident buffers without a copy.
"""

import ctypes as ct

import numpy as np

from . import _lib


class DeviceArray:
    __slots__ = ("ptr", "shape", "dtype", "_base", "_owned")

    def __init__(self, shape, dtype, ptr=None, base=None):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._base = base
        if ptr is None:
            p = ct.c_void_p()
            _lib.check(_lib.load().caf_malloc(ct.byref(p), self.nbytes), "caf_malloc")
            self.ptr = p.value or 0
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    def __del__(self):
        if getattr(self, "_owned", False) and self.ptr:
            try:
                _lib.load().caf_free(ct.c_void_p(self.ptr))
            except Exception:
                pass
            self.ptr = 0

    # -- numpy-like surface -------------------------------------------------------------
    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def itemsize(self):
        return self.dtype.itemsize

    def __len__(self):
        return self.shape[0]

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            _lib.check(_lib.load().caf_d2h(out.ctypes.data, ct.c_void_p(self.ptr), out.nbytes, None), "caf_d2h")
        return out

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.shape != self.shape:
            raise ValueError("shape mismatch: %s vs %s" % (host.shape, self.shape))
        if host.nbytes:
            _lib.check(_lib.load().caf_h2d(ct.c_void_p(self.ptr), host.ctypes.data, host.nbytes, None), "caf_h2d")
        return self

    def reshape(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)):
            shape = tuple(shape[0])
        shape = list(int(s) for s in shape)
        if -1 in shape:
            known = 1
            for s in shape:
                if s != -1:
                    known *= s
            shape[shape.index(-1)] = self.size // max(known, 1)
        v = DeviceArray(shape, self.dtype, ptr=self.ptr, base=self)
        if v.size != self.size:
            raise ValueError("cannot reshape array of size %d into shape %s" % (self.size, tuple(shape)))
        return v

    def __getitem__(self, key):
        """Views along the first axis only (an int row or a unit-step slice): what the reference's
        code paths use on device results (out[i], d_x[a:b])."""
        n0 = self.shape[0]
        row = self.dtype.itemsize
        for s in self.shape[1:]:
            row *= s
        if isinstance(key, (int, np.integer)):
            k = int(key)
            if k < 0:
                k += n0
            if not 0 <= k < n0:
                raise IndexError("index out of range")
            return DeviceArray(self.shape[1:], self.dtype, ptr=self.ptr + k * row, base=self)
        if isinstance(key, slice):
            a, b, st = key.indices(n0)
            if st != 1:
                raise ValueError("DeviceArray supports unit-step slices only")
            b = max(a, b)
            return DeviceArray((b - a,) + self.shape[1:], self.dtype, ptr=self.ptr + a * row, base=self)
        raise TypeError("DeviceArray supports int and slice indices on the first axis only")

    def copy(self):
        out = DeviceArray(self.shape, self.dtype)
        if self.nbytes:
            _lib.check(_lib.load().caf_d2d(ct.c_void_p(out.ptr), ct.c_void_p(self.ptr), self.nbytes, None), "caf_d2d")
            _lib.check(_lib.load().caf_stream_sync(None), "sync")
        return out

    def conj(self):
        # setup-time convenience (templates are small): round-trips through the host
        return asarray(self.get().conj())

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, ptr=0x%x)" % (self.shape, self.dtype, self.ptr)


def asarray(host, dtype=None):
    """Upload a host array (the cp.asarray of xcorrRoutines.py:81-82)."""
    if isinstance(host, DeviceArray):
        return host
    host = np.ascontiguousarray(host, dtype=dtype)
    return DeviceArray(host.shape, host.dtype).set(host)


def asnumpy(arr):
    return arr.get() if isinstance(arr, DeviceArray) else np.asarray(arr)


def empty(shape, dtype=np.float32):
    return DeviceArray(shape, dtype)


def zeros(shape, dtype=np.float32):
    a = DeviceArray(shape, dtype)
    if a.nbytes:
        _lib.check(_lib.load().caf_memset(ct.c_void_p(a.ptr), 0, a.nbytes, None), "caf_memset")
    return a


def free_all_blocks():
    """Return the allocator's cached blocks to the driver (cp.get_default_memory_pool().free_all_blocks())."""
    _lib.check(_lib.load().caf_pool_trim(), "caf_pool_trim")


def pool_stats():
    """dict(cached_bytes, in_use_bytes, hits, misses) of the caching allocator."""
    v = [ct.c_int64() for _ in range(4)]
    _lib.check(_lib.load().caf_pool_stats(*[ct.byref(x) for x in v]), "caf_pool_stats")
    return dict(zip(("cached_bytes", "in_use_bytes", "hits", "misses"), (int(x.value) for x in v)))


def requireDeviceArray(var):
    """The reference's requireCupyArray (cupyHelpers.py:74-77): TypeError for host arrays."""
    if not isinstance(var, DeviceArray):
        raise TypeError("Must be a device array (pydsproutines_amd.devarray.DeviceArray).")


def requireDtype(dtype, var):
    """The reference's cupyRequireDtype (cupyHelpers.py:50-69)."""
    if isinstance(dtype, (list, tuple)):
        if not any(var.dtype == np.dtype(d) for d in dtype):
            raise TypeError("Must be one of %s, found %s" % (", ".join(str(np.dtype(d)) for d in dtype), var.dtype))
    elif var.dtype != np.dtype(dtype):
        raise TypeError("Must be %s, found %s" % (np.dtype(dtype), var.dtype))


def as_device_ptr(obj):
    """(pointer, nbytes) of a DeviceArray or of a torch CUDA tensor."""
    if isinstance(obj, DeviceArray):
        return obj.ptr, obj.nbytes
    if hasattr(obj, "data_ptr") and hasattr(obj, "is_cuda"):
        if not obj.is_cuda:
            raise TypeError("torch tensor must live on the GPU")
        if not obj.is_contiguous():
            raise ValueError("torch tensor must be contiguous")
        return int(obj.data_ptr()), int(obj.numel() * obj.element_size())
    raise TypeError("expected a DeviceArray or a CUDA torch tensor, got %r" % type(obj))
