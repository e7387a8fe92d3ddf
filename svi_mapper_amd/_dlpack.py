"""torch tensors that ALIAS device memory owned by the library, through a DLPack capsule built with ctypes.

`torch.as_tensor(obj_with___cuda_array_interface__)` asks the runtime which device a pointer belongs to
(hipPointerGetAttributes) and trusts the ordinal it gets; on hosts where the process sees one GPU of several that lookup has
been observed to come back with an ordinal torch cannot select ("HIP error: invalid device ordinal" at the first use of the
tensor, intermittently, depending on the box).  A DLPack capsule states the device itself, so nothing is looked up.
"""
import ctypes as C

import torch

_kDLCPU, _kDLROCM = 1, 10
_CODES = {torch.float32: (2, 32), torch.float64: (2, 64), torch.int32: (0, 32), torch.int64: (0, 64), torch.uint8: (1, 8),
          torch.int8: (0, 8)}


class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int32), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]

_live = {}   # address of the managed tensor -> the ctypes objects it points into (kept until torch calls the deleter)


@_DELETER
def _deleter(p):
    _live.pop(C.addressof(p.contents), None)


_capsule_new = C.pythonapi.PyCapsule_New
_capsule_new.restype = C.py_object
_capsule_new.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]


def alias(ptr, shape, dtype, device):
    """A tensor of `shape` / `dtype` on `device` (torch.device or index; -1 / "cpu" for host memory) over the memory at `ptr`.
    The memory stays the caller's: it must outlive every use of the tensor."""
    shape = tuple(int(x) for x in shape)
    dev = torch.device(device) if not isinstance(device, int) else (torch.device("cuda", device) if device >= 0 else torch.device("cpu"))
    n = 1
    for x in shape:
        n *= x
    if n == 0 or not ptr:
        return torch.empty(shape, dtype=dtype, device=dev)
    code, bits = _CODES[dtype]
    shp = (C.c_int64 * max(len(shape), 1))(*shape)
    m = _DLManagedTensor()
    m.dl_tensor.data = C.c_void_p(int(ptr))
    if dev.type == "cpu":
        m.dl_tensor.device = _DLDevice(_kDLCPU, 0)
    else:
        m.dl_tensor.device = _DLDevice(_kDLROCM, dev.index if dev.index is not None else torch.cuda.current_device())
    m.dl_tensor.ndim = len(shape)
    m.dl_tensor.dtype = _DLDataType(code, bits, 1)
    m.dl_tensor.shape = C.cast(shp, C.POINTER(C.c_int64))
    m.dl_tensor.strides = None
    m.dl_tensor.byte_offset = 0
    m.manager_ctx = None
    m.deleter = _deleter
    _live[C.addressof(m)] = (m, shp)
    return torch.utils.dlpack.from_dlpack(_capsule_new(C.addressof(m), b"dltensor", None))
