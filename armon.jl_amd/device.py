"""Device object and device array type of the HIP backend.

Counterpart of the backend hooks of the reference: ``create_device(::Val{:ROCM})`` /
``device_array_type`` / ``device_memory_info`` / ``Base.wait`` (ref src/parameters.jl:751-802,921-951,
1031-1038; ext/ArmonAMDGPU.jl:9-27). The arrays are raw HBM allocations made through the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class HIPDevice:
    """One context = one GPU + one stream (ref ``params.device``)."""

    def __init__(self, device_id=0, stream=None):
        L = _lib.lib()
        ctx = C.c_void_p()
        check(L.armon_hip_init(int(device_id), C.c_void_p(stream) if stream else None, C.byref(ctx)))
        self._L = L
        self.ctx = ctx
        self.device_id = int(device_id)
        self.owns_ctx = True

    @classmethod
    def adopt(cls, ctx, device_id=0):
        """Wrap an existing ``armon_ctx*`` (a tile context owned by an ``armon_mgpu`` group): never destroyed here."""
        d = cls.__new__(cls)
        d._L = _lib.lib()
        d.ctx = ctx if isinstance(ctx, C.c_void_p) else C.c_void_p(ctx)
        d.device_id = int(device_id)
        d.owns_ctx = False
        return d

    def close(self):
        if self.ctx:
            if self.owns_ctx:
                self._L.armon_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def wait(self):
        """``Base.wait(params)``"""
        check(self._L.armon_hip_sync(self.ctx))

    def memory_info(self):
        free, total = C.c_size_t(), C.c_size_t()
        check(self._L.armon_hip_device_memory_info(self.ctx, C.byref(free), C.byref(total)))
        return free.value, total.value

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        check(self._L.armon_hip_device_name(self.ctx, buf, 256))
        return buf.value.decode()

    @property
    def stream(self):
        return self._L.armon_hip_stream(self.ctx)

    def timer_start(self):
        check(self._L.armon_hip_timer_start(self.ctx))

    def timer_stop(self):
        ms = C.c_double()
        check(self._L.armon_hip_timer_stop(self.ctx, C.byref(ms)))
        return ms.value

    def stream_copy4(self, src, dst, nbytes):
        """Measurement aid: copy 4 device arrays into 4 others in one launch (the sweep's traffic shape)."""
        a = (C.c_void_p * 4)(*[x.ptr for x in src])
        b = (C.c_void_p * 4)(*[x.ptr for x in dst])
        check(self._L.armon_hip_stream_copy4(self.ctx, C.byref(a), C.byref(b), int(nbytes)))

    def event_sync(self, slot):
        check(self._L.armon_hip_event_sync(self.ctx, int(slot)))

    def get_tuning(self, knob):
        v = C.c_int()
        check(self._L.armon_hip_get_tuning(self.ctx, knob.encode(), C.byref(v)))
        return v.value

    def y_run_rows(self):
        """Rows per run of this context's last Y sweep (armon_hip_get_tuning "Y_RUN_ROWS")."""
        return self.get_tuning("Y_RUN_ROWS")

    def pinned(self, n, dtype=np.float64):
        """``n`` elements of page-locked host memory as a numpy array (freed with the returned object)."""
        return PinnedArray(self, n, dtype)

    def event_record(self, slot):
        check(self._L.armon_hip_event_record(self.ctx, int(slot)))

    def event_elapsed_ms(self, a, b):
        ms = C.c_double()
        check(self._L.armon_hip_event_elapsed_ms(self.ctx, int(a), int(b), C.byref(ms)))
        return ms.value

    # array constructors ------------------------------------------------------------------------
    def empty(self, n, dtype=np.float64):
        return DeviceArray(self, n, dtype)

    def zeros(self, n, dtype=np.float64):
        a = DeviceArray(self, n, dtype)
        a.fill_bytes(0)
        return a

    def from_host(self, host):
        host = np.ascontiguousarray(host)
        a = DeviceArray(self, host.size, host.dtype)
        a.copy_from_host(host)
        return a


class DeviceArray:
    """Flat 1-D device vector: the ``V{T,1}(undef, n)`` of ref src/blocking/blocks.jl:36-44."""

    def __init__(self, device, n, dtype=np.float64):
        self.device = device
        self.n = int(n)
        self.dtype = np.dtype(dtype)
        self.nbytes = self.n * self.dtype.itemsize
        p = C.c_void_p()
        check(device._L.armon_hip_malloc(device.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    @classmethod
    def view(cls, owner, byte_offset, n, dtype):
        """A sub-array of ``owner``'s allocation (no ownership: it keeps ``owner`` alive instead)."""
        a = cls.__new__(cls)
        a.device, a.n, a.dtype = owner.device, int(n), np.dtype(dtype)
        a.nbytes = a.n * a.dtype.itemsize
        assert byte_offset % 256 == 0 and byte_offset + a.nbytes <= owner.nbytes
        a.ptr = owner.ptr + byte_offset
        a.owner = owner
        return a

    def __len__(self):
        return self.n

    @property
    def __cuda_array_interface__(self):
        """Zero-copy view for torch (``torch.as_tensor(a, device="cuda")``): collectives on device scalars."""
        return {"shape": (self.n,), "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2}

    def free(self):
        if getattr(self, "owner", None) is not None:
            self.ptr, self.owner = 0, None
            return
        if self.ptr and self.device.ctx:
            self.device._L.armon_hip_free(self.device.ctx, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def fill_bytes(self, byte):
        check(self.device._L.armon_hip_memset(self.device.ctx, C.c_void_p(self.ptr), byte, self.nbytes))

    def copy_from_host(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.n, (host.size, self.n)
        check(self.device._L.armon_hip_memcpy(self.device.ctx, C.c_void_p(self.ptr),
                                              host.ctypes.data_as(C.c_void_p), self.nbytes,
                                              _lib_kind("H2D")))

    def to_host(self, out=None):
        if out is None:
            out = np.empty(self.n, dtype=self.dtype)
        assert out.size == self.n and out.dtype == self.dtype and out.flags["C_CONTIGUOUS"]
        check(self.device._L.armon_hip_memcpy(self.device.ctx, out.ctypes.data_as(C.c_void_p),
                                              C.c_void_p(self.ptr), self.nbytes, _lib_kind("D2H")))
        return out

    def copy_from_device(self, other):
        assert other.nbytes == self.nbytes
        check(self.device._L.armon_hip_memcpy(self.device.ctx, C.c_void_p(self.ptr),
                                              C.c_void_p(other.ptr), self.nbytes, _lib_kind("D2D")))


def _lib_kind(k):
    return {"H2D": 1, "D2H": 2, "D2D": 3}[k]


class PinnedArray:
    """Page-locked host vector: the landing zone of asynchronous device → host copies."""

    def __init__(self, device, n, dtype=np.float64):
        self.device, self.dtype = device, np.dtype(dtype)
        self.nbytes = int(n) * self.dtype.itemsize
        p = C.c_void_p()
        check(device._L.armon_hip_malloc_host(device.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self.ptr), dtype=self.dtype)

    def copy_from_device_async(self, src, n=None, dst_offset=0, src_offset=0):
        """Enqueue a D2H copy of ``n`` elements on the context's stream; the caller orders it with an event."""
        n = len(src) - src_offset if n is None else n
        sz = self.dtype.itemsize
        check(self.device._L.armon_hip_memcpy_async(self.device.ctx, C.c_void_p(self.ptr + dst_offset * sz),
                                                    C.c_void_p(src.ptr + src_offset * sz), n * sz, _lib_kind("D2H")))

    def free(self):
        if self.ptr and self.device.ctx:
            self.array = None
            self.device._L.armon_hip_free_host(self.device.ctx, C.c_void_p(self.ptr))
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
