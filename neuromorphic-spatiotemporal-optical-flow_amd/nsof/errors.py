"""Error convention of the boundary.

cv2 raises ``cv2.error`` when a CV_Assert of calcOpticalFlowFarneback fails (sizes differ,
channels != 1, pyr_scale >= 1).  nsof raises ``nsof.error`` -- a ValueError subclass for bad
arguments, a RuntimeError subclass for device failures -- and, when cv2 is importable,
``nsof.error`` also derives from ``cv2.error`` so existing ``except cv2.error`` blocks keep working.
"""
from . import _lib

try:  # pragma: no cover - cv2 is absent in the build image
    import cv2 as _cv2
    _bases = (_cv2.error,)
except Exception:  # noqa: BLE001
    _bases = ()


class NsofError(*(_bases or (Exception,))):
    """Base class; ``.status`` holds the nsof_status code."""

    def __init__(self, msg, status=_lib.NSOF_EINVAL):
        super().__init__(msg)
        self.status = status


class NsofValueError(NsofError, ValueError):
    pass


class NsofDeviceError(NsofError, RuntimeError):
    pass


error = NsofError  # cv2-style alias


def raise_for_status(rc, ctx_ptr=None, what=""):
    if rc == _lib.NSOF_OK:
        return
    msg = _lib.load().nsof_last_error(ctx_ptr)
    msg = msg.decode("utf-8", "replace") if msg else ""
    text = f"{what}: {msg} (nsof_status {rc})" if what else f"{msg} (nsof_status {rc})"
    if rc in (_lib.NSOF_EDEVICE, _lib.NSOF_ENOMEM):
        raise NsofDeviceError(text, rc)
    raise NsofValueError(text, rc)
