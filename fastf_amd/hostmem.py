"""Host <-> device transfers of the Python layer (dist.py, workload.py, bench.py, the tests) through PINNED staging.

Pageable host memory (a numpy array, a torch CPU tensor) is never handed to the HIP runtime by this repo's code: for pageable
memory of a megabyte or more the runtime pins the caller's pages on the fly and has the GPU read or write them in place, and
all three GPU memory faults of rounds 5/6 were GPU writes to heap addresses while such a copy ran.  Here
every transfer goes through one grow-only pinned buffer (hipHostMalloc through torch's pinned allocator: memory the runtime
owns and whose lifetime it knows), in chunks, synchronously.  Not a hot path: the product's hot paths take pinned or device
memory through the C ABI.
"""
import numpy as np
import torch

_CHUNK = 32 << 20
_stage = None


def _staging(nbytes):
    global _stage
    need = min(max(int(nbytes), 1), _CHUNK)
    if _stage is None or _stage.numel() < need:
        _stage = None                                   # (give the old block back to the pinned allocator first)
        _stage = torch.empty(max(need, 1 << 20), dtype=torch.uint8, pin_memory=True)
    return _stage


def to_device(arr, device, dtype=None):
    """numpy array -> new device tensor.  dtype: a torch dtype of the same item size to VIEW the bytes as (torch has no
    uint64 / uint32 arithmetic: the callers carry those as int64 / int32 bit patterns)."""
    a = np.ascontiguousarray(arr)
    if torch.device(device).type == "cpu":
        signed = {np.dtype(np.uint64): np.int64, np.dtype(np.uint32): np.int32, np.dtype(np.uint16): np.int16}.get(a.dtype)
        t = torch.from_numpy((a.view(signed) if signed else a).copy())
        return t if dtype is None or t.dtype == dtype else t.view(dtype)
    if dtype is None:
        dtype = {np.dtype(np.uint64): torch.int64, np.dtype(np.uint32): torch.int32, np.dtype(np.uint16): torch.int16}.get(a.dtype)
        if dtype is None:
            dtype = torch.from_numpy(np.zeros(0, a.dtype)).dtype
    out = torch.empty(a.shape, dtype=dtype, device=device)
    nbytes = a.nbytes
    if nbytes == 0:
        return out
    assert out.numel() * out.element_size() == nbytes, "dtype view must keep the item size"
    src = a.reshape(-1).view(np.uint8)
    dst = out.reshape(-1).view(torch.uint8)
    st = _staging(nbytes)
    st_np = st.numpy()
    for off in range(0, nbytes, _CHUNK):
        n = min(_CHUNK, nbytes - off)
        st_np[:n] = src[off:off + n]
        dst[off:off + n].copy_(st[:n], non_blocking=True)
        torch.cuda.current_stream(out.device).synchronize()      # the staging buffer is reused by the next chunk
    return out


def to_host(t):
    """device (or CPU) tensor -> new numpy array of the tensor's dtype"""
    if t.device.type == "cpu":
        return t.numpy().copy()
    tc = t.contiguous()
    proto = torch.empty(0, dtype=tc.dtype).numpy().dtype
    out = np.empty(tuple(tc.shape), dtype=proto)
    nbytes = out.nbytes
    if nbytes == 0:
        return out
    src = tc.reshape(-1).view(torch.uint8)
    dst = out.reshape(-1).view(np.uint8)
    st = _staging(nbytes)
    st_np = st.numpy()
    for off in range(0, nbytes, _CHUNK):
        n = min(_CHUNK, nbytes - off)
        st[:n].copy_(src[off:off + n])                           # device -> pinned: synchronous for the host
        dst[off:off + n] = st_np[:n]
    return out


def copy_into(dst, src):
    """dst: a device (or CPU) tensor; src: a CPU tensor or numpy array of the same byte size -> dst's bytes"""
    a = src.numpy() if isinstance(src, torch.Tensor) else np.ascontiguousarray(src)
    if dst.device.type == "cpu":
        dst.reshape(-1).view(torch.uint8).copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1).view(np.uint8)))
        return dst
    d = dst.reshape(-1).view(torch.uint8) if dst.is_contiguous() else None
    if d is None:
        dst.copy_(to_device(a, dst.device, dst.dtype).reshape(dst.shape))
        return dst
    srcb = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    nbytes = srcb.nbytes
    assert nbytes == d.numel()
    if nbytes == 0:
        return dst
    st = _staging(nbytes)
    st_np = st.numpy()
    for off in range(0, nbytes, _CHUNK):
        n = min(_CHUNK, nbytes - off)
        st_np[:n] = srcb[off:off + n]
        d[off:off + n].copy_(st[:n], non_blocking=True)
        torch.cuda.current_stream(dst.device).synchronize()
    return dst


def to_host_tensor(t):
    """device tensor -> CPU tensor (for the host-staged collectives of dist.py)"""
    return torch.from_numpy(to_host(t))
