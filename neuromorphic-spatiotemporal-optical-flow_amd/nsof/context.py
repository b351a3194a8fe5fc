"""Device context: one GPU, one HIP stream, one reusable workspace (nsof_ctx)."""
import ctypes as C
import os
import threading

from . import _lib
from .errors import raise_for_status


def dev_ptr(obj):
    """Device address of a torch tensor / anything exposing data_ptr() or __cuda_array_interface__."""
    if obj is None:
        return None
    if isinstance(obj, int):
        return obj
    if hasattr(obj, "data_ptr"):
        return obj.data_ptr()
    if hasattr(obj, "__cuda_array_interface__"):
        return obj.__cuda_array_interface__["data"][0]
    raise TypeError(f"cannot take a device pointer of {type(obj)!r}")


class Context:
    """Owns an ``nsof_ctx``.  One per thread / per GPU is the intended use; ``lock`` (re-entrant) serialises the entry points
    that change options around a call (``calcOpticalFlowFarneback(exact=..., low_latency=...)``) for callers that share
    one context, e.g. the process-wide ``default_context()``."""

    def __init__(self, device=None):
        self._lib = _lib.load()
        if device is None:
            device = int(os.environ.get("NSOF_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.device = int(device)
        p = C.c_void_p()
        rc = self._lib.nsof_create(self.device, C.byref(p))
        raise_for_status(rc, None, "nsof_create")
        self._p = p
        self.lock = threading.RLock()

    # -- plumbing -------------------------------------------------------------------------
    @property
    def ptr(self):
        if self._p is None:
            raise RuntimeError("nsof context already destroyed")
        return self._p

    def check(self, rc, what=""):
        raise_for_status(rc, self._p, what)

    def set_stream(self, stream):
        """Launch on a caller-owned HIP stream (``torch.cuda.current_stream().cuda_stream``); None = own."""
        self.check(self._lib.nsof_set_stream(self.ptr, C.c_void_p(stream) if stream else None), "set_stream")

    def set_option(self, option, value):
        """``nsof_set_option``: e.g. ``ctx.set_option(_lib.OPT_POLYEXP_F32, 1)`` (float polynomial expansion, opt-in)."""
        self.check(self._lib.nsof_set_option(self.ptr, int(option), int(value)), "set_option")

    def get_option(self, option):
        v = C.c_int()
        self.check(self._lib.nsof_get_option(self.ptr, int(option), C.byref(v)), "get_option")
        return v.value

    def synchronize(self):
        self.check(self._lib.nsof_synchronize(self.ptr), "synchronize")

    def prof_enable(self, *kernel_ids):
        mask = 0
        for k in kernel_ids:
            mask |= 1 << k
        self.check(self._lib.nsof_prof_enable(self.ptr, mask), "prof_enable")

    def prof_collect(self, kernel_id):
        """-> (total_ms, launches) of the bracketed kernel since the last collect (synchronises)."""
        ms, n = C.c_double(), C.c_longlong()
        self.check(self._lib.nsof_prof_collect(self.ptr, kernel_id, C.byref(ms), C.byref(n)), "prof_collect")
        return ms.value, n.value

    def close(self):
        if getattr(self, "_p", None) is not None:
            self._lib.nsof_destroy(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_default = None
_lock = threading.Lock()


def default_context():
    """Process-wide lazily created context used by the cv2-style entry points."""
    global _default
    with _lock:
        if _default is None:
            _default = Context()
        return _default
