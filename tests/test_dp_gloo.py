"""World-size-2 and -8 data parallelism on CPU (gloo): shard utterances, all-reduce the UNNORMALISED flat
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


def _problem(B=6):
    T, F, r, K = 7, 21, 5, 2
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
    torch.set_num_threads(1)
    P, alt, labels, K, kc, kn, w = _problem(_SEQS[world])
    lo, hi = dp.shard(P["X"].shape[0])
    flat = _flat_grads(P, alt, labels, K, kc, kn, w, lo, hi)
    dp.allreduce_sum_(flat)
    g, loss = dp.normalised(flat, n_scalars=2)      # this test's buffer ends in [sum, count]
    np.save(os.path.join(out_dir, "g%d.npy" % rank), np.concatenate([g.numpy(), [loss]]))
    torch.distributed.destroy_process_group()


def test_normalised_reads_the_models_scalar_tail():
    # the model's flat buffer ends in [sum w*mse, count, rows, fault] (layers.UnfoldedSNMFModel.N_SCALARS
    # = 4): the default must index the tail from its START (rows is not the count)
    from drnmf_amd import dp, layers
    assert dp.N_SCALARS == layers.UnfoldedSNMFModel.N_SCALARS == 4
    flat = torch.tensor([2.0, 4.0, 6.0, 10.0, 4.0, 64.0, 0.0], dtype=torch.float64)
    g, loss = dp.normalised(flat)
    np.testing.assert_allclose(g.numpy(), [0.5, 1.0, 1.5])
    assert loss == 2.5
    g2, loss2 = dp.normalised(flat[:5], n_scalars=2)
    np.testing.assert_allclose(g2.numpy(), [0.5, 1.0, 1.5])
    assert loss2 == 2.5


def test_shard_covers_everything():
    from drnmf_amd import dp
    for n in (1, 5, 8, 13):
        for w in (1, 2, 3, 8):
            spans = [dp.shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


# sequences per world size: 6 over 2 ranks (3 + 3); 10 over 8 ranks (2, 2, 1, 1, 1, 1, 1, 1 -- uneven, six
# ranks with a single sequence: BASELINE configs[3] is 8 ranks)
_SEQS = {2: 6, 8: 10}


@pytest.mark.parametrize("world", [2, 8])
def test_allreduce_equals_unsharded(tmp_path, world):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = [np.load(tmp_path / ("g%d.npy" % r)) for r in range(world)]
    g0 = g[0]
    for gr in g[1:]:
        np.testing.assert_array_equal(g0, gr)              # every rank applies the same step
    P, alt, labels, K, kc, kn, w = _problem(_SEQS[world])
    full = _flat_grads(P, alt, labels, K, kc, kn, w, 0, P["X"].shape[0])
    cnt = float(full[-1])
    ref = np.concatenate([(full[:-2] / cnt).numpy(), [float(full[-2]) / cnt]])
    np.testing.assert_allclose(g0, ref, rtol=1e-12, atol=1e-15)
    if world != 2:
        return
    # and it differs from the naive average of per-rank normalised gradients (ragged lengths)
    lo, hi = 0, 3
    a = _flat_grads(P, alt, labels, K, kc, kn, w, 0, 3)
    b = _flat_grads(P, alt, labels, K, kc, kn, w, 3, 6)
    naive = 0.5 * (a[:-2] / a[-1] + b[:-2] / b[-1]).numpy()
    assert np.max(np.abs(naive - ref[:-1])) > 1e-6 * np.max(np.abs(ref[:-1]))


def _plan_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from drnmf_amd import dp, layers
    bs = 8 if world == 2 else 2
    lo, hi = dp.shard(17)        # 2 ranks: 9 / 8 sequences = 2 / 1 mini-batches of 8;
    n = hi - lo                  # 8 ranks: 3, 2, 2, ... sequences = 2 / 1 / 1 ... mini-batches of 2
    steps = dp.max_over_ranks(layers.epoch_steps(n, bs))
    plan = [(b.tolist(), live) for b, live in layers.epoch_batches(np.arange(n), bs, steps)]
    # every step is a collective in fit(): stand in for it with the flat-buffer all-reduce
    tot = []
    for b, live in plan:
        flat = torch.tensor([float(len(b)) if live else 0.0, 1.0], dtype=torch.float64)
        dp.allreduce_sum_(flat)
        tot.append(flat.tolist())
    # broadcast: rank 1 starts with different "weights"
    wts = torch.full((5,), float(rank + 1))
    dp.broadcast_(wts, 0)
    np.save(os.path.join(out_dir, "plan%d.npy" % rank),
            np.array([steps, len(plan)] + [x for t in tot for x in t] + wts.tolist()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_uneven_shards_take_the_same_number_of_steps(tmp_path, world):
    """fit() under data parallelism: 17 utterances over 2 ranks with batch_size 8 give 2 and 1 local
    mini-batches (over 8 ranks with batch_size 2: 2 on rank 0, 1 on the seven others); every rank must
    run 2 collective steps (the short ranks join the second with zero weights), and broadcast_ makes
    the replicas equal."""
    port = _free_port()
    mp.spawn(_plan_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    plans = [np.load(tmp_path / ("plan%d.npy" % r)) for r in range(world)]
    p0 = plans[0]
    assert p0[0] == 2 and p0[1] == 2
    # step 0: every rank's first mini-batch is live (16 sequences either way), step 1: one live
    # sequence on rank 0, none elsewhere; all ranks took part in both reductions
    np.testing.assert_array_equal(p0[2:6], [16.0, float(world), 1.0, float(world)])
    for pr in plans[1:]:
        np.testing.assert_array_equal(p0, pr)
    np.testing.assert_array_equal(p0[6:], np.ones(5))


def test_epoch_batches_single_rank():
    from drnmf_amd import layers
    idx = np.arange(10)[::-1]
    got = list(layers.epoch_batches(idx, 4, layers.epoch_steps(10, 4)))
    assert [b.tolist() for b, _ in got] == [[9, 8, 7, 6], [5, 4, 3, 2], [1, 0]]
    assert all(live for _, live in got)
