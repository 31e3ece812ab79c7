"""World-size-2 data parallelism on CPU (gloo): shard utterances, all-reduce the UNNORMALISED flat
gradient + (sum, count), normalise after the reduce -- must equal the unsharded gradient even when
the ranks hold different numbers of valid frames.  Compute on each rank is
the torch oracle (test infrastructure); the code under test is drnmf_amd/dp.py, the same helper
the GPU training path and bench.py use with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import drnmf_oracle as O
from oracle import drnmf_torch_ref as TR


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    B, T, F, r, K = 6, 7, 21, 5, 2
    P = O.synth_problem(B, T, F, r, seed=9, ragged=True, density=0.2)
    N = 2 * r
    params = dict(W=P["W"], U1=np.eye(N, dtype=np.float32), Uk=np.zeros((N, N), np.float32),
                  alph=np.float32(N / 4.0), lam1=np.float32(0.3))
    alt, labels = O.build_alt(N, K, params, ("log_D", "log_alph"))
    kc = np.log(1e-7 + P["W"][:, :r]).T
    kn = np.log(1e-7 + P["W"][:, r:]).T
    w = (P["X"] != -1.0).any(-1).astype(np.float64)
    return P, alt, labels, K, kc, kn, w


def _flat_grads(P, alt, labels, K, kc, kn, w, lo, hi):
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    leaves = {k: t64(v).requires_grad_(True) for k, v in alt.items() if k.startswith("log_D")}
    a = {k: (leaves[k] if k in leaves else t64(v)) for k, v in alt.items()}
    lh0 = t64(P["log_h0"]).requires_grad_(True)
    loss, _, _ = TR.model_loss(t64(P["X"][lo:hi]), t64(P["Y"][lo:hi]), t64(w[lo:hi]), a, labels, K,
                               lh0, t64(kc), t64(kn), normalise=False)
    loss.backward()
    g = [leaves[k].grad.reshape(-1) for k in sorted(leaves)] + [lh0.grad.reshape(-1)]
    cnt = float((w[lo:hi] != 0).sum())
    return torch.cat(g + [loss.detach().reshape(1), torch.tensor([cnt], dtype=torch.float64)])


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from drnmf_amd import dp
    P, alt, labels, K, kc, kn, w = _problem()
    lo, hi = dp.shard(P["X"].shape[0])
    flat = _flat_grads(P, alt, labels, K, kc, kn, w, lo, hi)
    dp.allreduce_sum_(flat)
    g, loss = dp.normalised(flat)
    np.save(os.path.join(out_dir, "g%d.npy" % rank), np.concatenate([g.numpy(), [loss]]))
    torch.distributed.destroy_process_group()


def test_shard_covers_everything():
    from drnmf_amd import dp
    for n in (1, 5, 8, 13):
        for w in (1, 2, 3, 8):
            spans = [dp.shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_two_rank_allreduce_equals_unsharded(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = np.load(tmp_path / "g0.npy")
    g1 = np.load(tmp_path / "g1.npy")
    np.testing.assert_array_equal(g0, g1)                  # every rank applies the same step
    P, alt, labels, K, kc, kn, w = _problem()
    full = _flat_grads(P, alt, labels, K, kc, kn, w, 0, P["X"].shape[0])
    cnt = float(full[-1])
    ref = np.concatenate([(full[:-2] / cnt).numpy(), [float(full[-2]) / cnt]])
    np.testing.assert_allclose(g0, ref, rtol=1e-12, atol=1e-15)
    # and it differs from the naive average of per-rank normalised gradients (ragged lengths)
    lo, hi = 0, 3
    a = _flat_grads(P, alt, labels, K, kc, kn, w, 0, 3)
    b = _flat_grads(P, alt, labels, K, kc, kn, w, 3, 6)
    naive = 0.5 * (a[:-2] / a[-1] + b[:-2] / b[-1]).numpy()
    assert np.max(np.abs(naive - ref[:-1])) > 1e-6 * np.max(np.abs(ref[:-1]))
