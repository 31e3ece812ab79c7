"""Data parallelism over utterances: one process per GPU, parameters replicated, ONE all-reduce per
optimiser step over a flat fp32 buffer [gradients..., sum w*mse, count, rows].  The loss
normalisation (1/count) is applied AFTER the reduce with the global count: Keras normalises by
batch-level statistics, so averaging per-rank normalised gradients is wrong when ranks hold
different numbers of valid frames.

The collective itself is libdrnmf's (include/drnmf.h: drnmf_comm_init / drnmf_allreduce_grads /
drnmf_broadcast_params, RCCL over xGMI, enqueued on the caller's stream): a non-Python host trains
on N GPUs through the same C entry points.  `torch.distributed` is the launcher-side rendezvous
only -- it tells the ranks who they are and carries the 128-byte RCCL id from rank 0 to the others.
CPU tensors (the gloo tests of the host logic) and DRNMF_DP_BACKEND=torch (several test ranks on ONE
GPU, which RCCL refuses) go through torch.distributed.all_reduce instead.
"""
import ctypes as C
import os

import torch


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def world_size():
    return torch.distributed.get_world_size() if is_distributed() else 1


def rank():
    return torch.distributed.get_rank() if is_distributed() else 0


_comms = {}      # device index -> (rank, world) of the communicator owned by that device's handle


def _use_abi(t):
    return t.is_cuda and os.environ.get("DRNMF_DP_BACKEND", "rccl") != "torch"


def comm_init(device, rank_=None, world_=None):
    """Create the RCCL communicator of `device`'s library handle (collective: every rank calls it).
    rank/world default to torch.distributed's; the id travels from rank 0 by broadcast_object_list.
    world == 1 without a process group makes a single-rank communicator (used by the GPU test of
    the ABI path on a one-GPU box)."""
    from . import _capi
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _comms:
        return _comms[idx]
    r = rank() if rank_ is None else int(rank_)
    w = world_size() if world_ is None else int(world_)
    L, h = _capi.lib(), _capi.handle(idx)
    buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
    err = None
    if r == 0:
        # (a failure here -- librccl not loadable -- must still reach the broadcast below: the other
        # ranks are waiting in it)
        try:
            _capi.check(L.drnmf_comm_unique_id(h, buf), h, "drnmf_comm_unique_id")
        except Exception as e:       # noqa: BLE001
            err = repr(e)
    if w > 1:
        box = [(buf.raw, err) if r == 0 else None]
        torch.distributed.broadcast_object_list(box, src=0)
        raw, err = box[0]
        buf = C.create_string_buffer(raw, _capi.COMM_ID_BYTES)
    if err is not None:
        raise RuntimeError("drnmf_comm_unique_id failed on rank 0: " + err)
    with torch.cuda.device(idx):
        _capi.check(L.drnmf_comm_init(h, buf, r, w), h, "drnmf_comm_init")
    _comms[idx] = (r, w)
    return _comms[idx]


def comm_info(device=None):
    """(rank, world) as the LIBRARY's communicator reports them (drnmf_comm_info), or None when the
    device's handle owns no communicator."""
    from . import _capi
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
    if idx not in _comms:
        return None
    r, w = C.c_int32(-1), C.c_int32(-1)
    h = _capi.handle(idx)
    _capi.check(_capi.lib().drnmf_comm_info(h, C.byref(r), C.byref(w)), h, "drnmf_comm_info")
    return int(r.value), int(w.value)


def comm_destroy(device=None):
    from . import _capi
    for idx in list(_comms) if device is None else [torch.device(device).index or 0]:
        if idx in _comms:
            h = _capi.handle(idx)
            _capi.check(_capi.lib().drnmf_comm_destroy(h), h, "drnmf_comm_destroy")
            del _comms[idx]


def _abi_call(name, t, *extra):
    from . import _capi
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("%s needs a contiguous float32 tensor" % name)
    idx = t.device.index if t.device.index is not None else torch.cuda.current_device()
    if idx not in _comms:
        comm_init(t.device)
    h = _capi.handle(idx)
    stream = C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    rc = getattr(_capi.lib(), name)(h, _capi.ptr(t), t.numel(), *extra, stream)
    _capi.check(rc, h, name)


def allreduce_sum_(flat, force=False):
    """In-place sum over ranks of a flat fp32 tensor; no-op for a single rank (unless `force`,
    which sends it through the single-rank communicator: the test of the ABI path)."""
    dev_comm = flat.is_cuda and (flat.device.index in _comms)
    if world_size() <= 1 and not (force and dev_comm):
        return flat
    if _use_abi(flat):
        _abi_call("drnmf_allreduce_grads", flat)
    else:
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    return flat


def broadcast_(flat, root=0, force=False):
    """Rank `root`'s buffer replaces every rank's (replica consistency at compile())."""
    dev_comm = flat.is_cuda and (flat.device.index in _comms)
    if world_size() <= 1 and not (force and dev_comm):
        return flat
    if _use_abi(flat):
        _abi_call("drnmf_broadcast_params", flat, int(root))
    else:
        torch.distributed.broadcast(flat, src=root)
    return flat


def max_over_ranks(v):
    """Largest value of a host integer over the ranks (fit(): steps per epoch).  Host-side control
    data: carried by the launcher's process group, not by the gradient communicator."""
    if world_size() <= 1:
        return int(v)
    box = [None] * world_size()
    torch.distributed.all_gather_object(box, int(v))
    return max(box)


def shard(n_items, r=None, w=None):
    """Contiguous shard [lo, hi) of n_items sequences for rank r of w (even split, remainder to the
    first ranks) -- utterances are independent in the forward and backward passes."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n_items, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


N_SCALARS = 4        # tail of the model's flat buffer: [sum w*mse, count, rows, fault] (layers.UnfoldedSNMFModel.N_SCALARS)


def normalised(flat, n_scalars=N_SCALARS):
    """(gradient / global count, loss) from an all-reduced flat buffer whose last `n_scalars`
    entries START with [sum w*mse, count] (the model's buffer carries the frame count and the fault word
    behind them)."""
    if n_scalars < 2:
        raise ValueError("the scalar tail holds at least [sum, count]")
    tail = flat[flat.numel() - n_scalars:]
    cnt = float(tail[1])
    scale = 1.0 / max(cnt, 1.0)
    return flat[:flat.numel() - n_scalars] * scale, float(tail[0]) * scale
