"""Data parallelism over utterances: one process per GPU, parameters replicated, ONE all-reduce per
optimiser step over a flat fp32 buffer [gradients..., sum w*mse, count] (RCCL over xGMI on GPUs;
gloo in the CPU tests).  The loss normalisation (1/count) is applied AFTER the reduce with the
global count: Keras normalises by batch-level statistics, so averaging per-rank normalised
gradients is wrong when ranks hold different numbers of valid frames."""
import torch


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def world_size():
    return torch.distributed.get_world_size() if is_distributed() else 1


def rank():
    return torch.distributed.get_rank() if is_distributed() else 0


def allreduce_sum_(flat):
    """In-place sum over ranks of a flat tensor; no-op without an initialised process group."""
    if is_distributed() and torch.distributed.get_world_size() > 1:
        torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    return flat


def shard(n_items, r=None, w=None):
    """Contiguous shard [lo, hi) of n_items sequences for rank r of w (even split, remainder to the
    first ranks) -- utterances are independent in the forward and backward passes."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n_items, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def normalised(flat):
    """(gradient / global count, loss) from an all-reduced flat buffer."""
    cnt = float(flat[-1])
    scale = 1.0 / max(cnt, 1.0)
    return flat[:-2] * scale, float(flat[-2]) * scale
