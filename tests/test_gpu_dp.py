"""Data parallelism of the PRODUCT training step on the GPU.

1. The collective behind the C ABI (drnmf_comm_init / drnmf_allreduce_grads / drnmf_broadcast_params,
   RCCL resolved by dlopen) on the one GPU of the test box: a single-rank communicator is a real
   RCCL communicator, so the whole call path (id, init, stream-ordered collectives, destroy) runs.
2. Two ranks sharing that GPU (RCCL refuses duplicate devices, so the gradient exchange goes through
   the gloo process group: DRNMF_DP_BACKEND=torch): each rank runs the product's own
   fit() / train_on_batch -- HIP forward, BPTT, Adam -- on ITS shard of 17 utterances (9 + 8, batch
   size 8: the short rank joins the last reduction with zero weights), with DIFFERENT initial
   log_h0 per rank (compile() must broadcast rank 0's) and a keras204 loss normalisation.
   Expected: both ranks end with bit-identical weights, equal (fp32 summation order aside) to ONE
   process stepping over the same global batches, and only rank 0 writes the checkpoint.
3. loss_norm='keras204' against the oracle's 'keras_mask_and_weight'.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import drnmf_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    from drnmf_amd import _capi
    _capi.handle(0)
    return torch.device("cuda:0")


def test_rccl_communicator_through_the_c_abi(dev):
    import ctypes as C
    from drnmf_amd import _capi, dp
    L, h = _capi.lib(), _capi.handle(0)
    r, w = C.c_int32(-1), C.c_int32(-1)
    assert L.drnmf_comm_info(h, C.byref(r), C.byref(w)) == 0 and w.value == 1
    flat = torch.arange(1, 1001, dtype=torch.float32, device=dev)
    # no communicator yet: a loud error, not a silent no-op
    rc = L.drnmf_allreduce_grads(h, _capi.ptr(flat), flat.numel(), None)
    assert rc == -1 and b"no communicator" in L.drnmf_last_error(h)
    assert dp.comm_init(dev, rank_=0, world_=1) == (0, 1)
    want = flat.clone()
    dp.allreduce_sum_(flat, force=True)
    dp.broadcast_(flat, 0, force=True)
    torch.cuda.synchronize()
    assert torch.equal(flat, want)                     # sum over one rank / broadcast from itself
    big = torch.rand(3_000_001, device=dev)            # the C3 gradient size class (12 MB)
    want = big.clone()
    dp.allreduce_sum_(big, force=True)
    torch.cuda.synchronize()
    assert torch.equal(big, want)
    rc = L.drnmf_broadcast_params(h, _capi.ptr(big), big.numel(), 3, None)
    assert rc == -1 and b"root" in L.drnmf_last_error(h)
    assert L.drnmf_comm_init(h, C.create_string_buffer(128), 0, 1) == -1   # already owns one
    dp.comm_destroy(dev)
    assert L.drnmf_comm_info(h, C.byref(r), C.byref(w)) == 0 and w.value == 1
    # the handle can own a communicator AGAIN (a second fit() / a re-formed group), and the collective takes
    # the headline's gradient buffer: 107 MB (configs[1] untied, K = 25: DESIGN.md section 5)
    assert dp.comm_init(dev, rank_=0, world_=1) == (0, 1)
    huge = torch.rand(26_750_000, device=dev)
    want = huge.clone()
    dp.allreduce_sum_(huge, force=True)
    dp.broadcast_(huge, 0, force=True)
    torch.cuda.synchronize()
    assert torch.equal(huge, want)
    dp.comm_destroy(dev)
    assert L.drnmf_comm_info(h, C.byref(r), C.byref(w)) == 0 and w.value == 1


def _session_matrix_mode():
    """Spawned workers do not run conftest's session fixture: take DRNMF_TEST_MATRIX_MODE themselves."""
    mode = os.environ.get("DRNMF_TEST_MATRIX_MODE")
    if mode:
        from drnmf_amd import ops
        ops.set_matrix_mode(mode, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


B_ALL, T, F, R, K = 17, 6, 33, 8, 3


def _problem():
    P = O.synth_problem(B_ALL, T, F, R, seed=21, ragged=True, density=0.15)
    w = (P["X"] != -1.0).any(-1).astype(np.float32)
    return P, w


def _model(seed):
    from drnmf_amd import layers
    P, _ = _problem()
    N = 2 * R
    np.random.seed(seed)            # log_h0 draws from the global generator: differs per rank
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=P["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
             params_trainable=["log_D", "log_alph"])
    return layers.build_unfolded_snmf(p, device="cuda:0")


def _dp_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DRNMF_DP_BACKEND"] = "torch"       # two ranks on one GPU: no RCCL
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    _session_matrix_mode()
    from drnmf_amd import callbacks, dp
    P, w = _problem()
    model = _model(seed=100 + rank)
    model.compile(lr=1e-2, loss_norm="keras204")
    lo, hi = dp.shard(B_ALL)
    ck = os.path.join(out_dir, "ck_rank%d.npz" % rank)
    hist = model.fit(P["X"][lo:hi], P["Y"][lo:hi], sample_weight=w[lo:hi], batch_size=8, epochs=2,
                     shuffle=False, validation_data=(P["X"][lo:hi], P["Y"][lo:hi], w[lo:hi]),
                     callbacks=[callbacks.ModelCheckpoint(ck, save_weights_only=True)])
    np.savez(os.path.join(out_dir, "w%d.npz" % rank), *model.get_weights(),
             loss=np.array(hist["loss"]), val=np.array(hist["val_loss"]))
    torch.distributed.destroy_process_group()


def _skipped_step_worker(rank, world, port, out_dir):
    """Twelve data-parallel steps with decay (lr_t depends on the step count); step 3 carries a fault word on
    rank 0 only (what drnmf_status_take_device leaves when a persistent chain of that rank timed out).  Rank 0
    reads that step's loss AT ONCE, rank 1 reads no loss at all -- it is told by a later train_on_batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DRNMF_DP_BACKEND"] = "torch"
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    _session_matrix_mode()
    from drnmf_amd import _capi, dp
    P, w = _problem()
    model = _model(seed=100 + rank)
    model.compile(lr=1e-2, decay=0.05)
    lo, hi = dp.shard(B_ALL)
    inner = model.loss_and_grads
    state = {"step": 0}

    def faulty(*a, **k):
        flat = inner(*a, **k)
        if state["step"] == 3 and rank == 0:
            flat[-1] += 1.0
        return flat
    model.loss_and_grads = faulty
    told = 0
    for step in range(12):
        state["step"] = step
        try:
            loss = model.train_on_batch(P["X"][lo:hi], P["Y"][lo:hi], w[lo:hi])
            if rank == 0 and step == 3:
                float(loss)
        except _capi.DrnmfError:
            told += 1
    torch.cuda.synchronize()
    assert told == 1, told                                  # every rank hears of the skipped step exactly once
    applied = float(model._step_dev[model._step_no & 1])
    np.savez(os.path.join(out_dir, "skip%d.npz" % rank), *model.get_weights(), applied=np.array([applied]),
             mirror=np.array([model.opt["iterations"]]))
    torch.distributed.destroy_process_group()


def test_skipped_step_keeps_the_replicas_identical_whoever_reads_the_loss(dev, tmp_path):
    """the bias correction / decay of Adam hang on the number of APPLIED steps.  That count
    lives on the device (drnmf_adam_step_flat_counted) and is not advanced by a step the all-reduced fault word
    skipped -- so the ranks of a data-parallel group keep passing the same lr_t whenever their host threads get
    round to reading the fault report (before: a host-side count, taken back when the report was read)."""
    port = _free_port()
    mp.spawn(_skipped_step_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "skip0.npz"), np.load(tmp_path / "skip1.npz")
    for k in [k for k in z0.files if k.startswith("arr_")]:
        np.testing.assert_array_equal(z0[k], z1[k])
    assert z0["applied"][0] == 11.0 and z1["applied"][0] == 11.0     # twelve enqueued, one skipped
    assert z0["mirror"][0] == 11 and z1["mirror"][0] == 11


def test_two_rank_product_training_equals_single_process(dev, tmp_path):
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "w0.npz"), np.load(tmp_path / "w1.npz")
    keys = [k for k in z0.files if k.startswith("arr_")]
    for k in keys:
        np.testing.assert_array_equal(z0[k], z1[k])         # replicas stay identical
    np.testing.assert_array_equal(z0["loss"], z1["loss"])
    np.testing.assert_array_equal(z0["val"], z1["val"])
    assert os.path.exists(tmp_path / "ck_rank0.npz") and not os.path.exists(tmp_path / "ck_rank1.npz")

    # one process, the same global batches: step 0 = sequences 0-7 (rank 0) + 9-16 (rank 1),
    # step 1 = sequence 8 (rank 1 joins the reduction with zeros), two epochs, rank 0's initial weights
    P, w = _problem()
    model = _model(seed=100)
    model.compile(lr=1e-2, loss_norm="keras204")
    g0 = list(range(0, 8)) + list(range(9, 17))
    losses = []
    for _ in range(2):
        a = model.train_on_batch(P["X"][g0], P["Y"][g0], w[g0])
        # rank 1's filler batch carries zero weights AND zero rows (p = valid / frames of
        # loss_norm='keras204' is the global batch's: sequence 8 alone)
        b = model.train_on_batch(P["X"][[8]], P["Y"][[8]], w[[8]])
        losses.append(0.5 * (a + b))
    for k, ref in zip(keys, model.get_weights()):
        got = z0[k]
        err = np.max(np.abs(got - ref)) / max(np.max(np.abs(ref)), 1e-30)
        assert err <= 2e-5, (k, err)
    np.testing.assert_allclose(z0["loss"], losses, rtol=2e-5)


def test_keras204_loss_normalisation_matches_oracle(dev):
    """loss_norm='keras204' = the oracle's 'keras_mask_and_weight' (weighted_masked_objective with
    a propagated mask, [K2.0.4-memory]); 'masked_mean' = its default.  Same masks, two scalings."""
    P, w = _problem()
    model = _model(seed=3)
    x = P["X"][:5]
    for norm, onorm in (("masked_mean", "masked_mean"), ("keras204", "keras_mask_and_weight")):
        model.compile(lr=1e-3, loss_norm=norm)
        mask = model.predict_on_batch(x)              # (the training step below moves the weights)
        got = model.test_on_batch(x, P["Y"][:5], w[:5])
        want = O.loss_mse_of_masked(x.astype(np.float64), mask.astype(np.float64),
                                    P["Y"][:5].astype(np.float64), w[:5], norm=onorm)
        assert abs(got - want) <= 1e-5 * abs(want), (norm, got, want)
        got_tr = model.train_on_batch(x, P["Y"][:5], w[:5])        # loss before the update
        assert abs(got_tr - want) <= 1e-5 * abs(want), (norm, got_tr, want)


def test_gram_and_factored_forms_agree(dev, monkeypatch):
    """csrc/cell_gram.h: the Gram form (one B x N x N contraction per layer-step, hoisted x Dn_k
    products, layer 0 folded into the first launch) against the factored kernels on the same
    inputs -- forward, every hidden layer, and the BPTT gradients; both within fp32 rounding of each
    other (they are checked against the fp64 oracle separately)."""
    from drnmf_amd import layers, ops
    P = O.synth_problem(37, 131, 65, 40, seed=8, ragged=True, density=0.1)   # > 2 ring blocks of frames
    N, K = 80, 4
    p = dict(input_dim=65, hidden_dim=N, output_dim=65, mask_value=-1., maxseq=131, K_layers=K,
             W=P["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
             params_trainable=["log_D", "log_alph", "log_lam1"])
    w = (P["X"] != -1.0).any(-1).astype(np.float32)
    res = {}
    for form in ("0", "1"):
        monkeypatch.setenv("DRNMF_GRAM", form)
        np.random.seed(5)
        model = layers.build_unfolded_snmf(p, device=dev)
        desc = model.cell._desc(37, 131)
        assert ops.cell_launches_per_frame(desc) == (K - 1 if form == "1" else 2 * K - 1)
        model.compile(lr=1e-3)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        flat = model.loss_and_grads(t(P["X"]), t(P["Y"]), t(w)).clone()
        hall = model.cell.forward_train(t(P["X"]), mask_value=-1.).clone()
        res[form] = (flat.cpu().numpy(), hall.cpu().numpy(), model.predict_on_batch(P["X"]))
    f0, h0, m0 = res["0"]
    f1, h1, m1 = res["1"]
    assert np.max(np.abs(h0 - h1)) <= 2e-5 * np.max(np.abs(h0))
    assert np.mean((m0 - m1) ** 2) <= 1e-12
    assert f0[-1] == f1[-1] == 0.0                                  # fault word
    assert f0[-3] == f1[-3] and f0[-2] == f1[-2] and abs(f0[-4] - f1[-4]) <= 1e-5 * abs(f0[-4])
    g0, g1 = f0[:-4], f1[:-4]
    assert np.max(np.abs(g0 - g1)) <= 2e-4 * np.max(np.abs(g0))


@pytest.mark.parametrize("cfg", [
    dict(B=1, T=150, F=65, r=20, K=4),           # single utterance, > 2 ring blocks of frames
    dict(B=7, T=70, F=33, r=50, K=3, ah=True),   # all-hidden (the training forward), ragged
    dict(B=16, T=9, F=129, r=100, K=2),          # K = 2: first and last layer-step in one phase
    dict(B=3, T=5, F=21, r=250, K=5),            # N = 500: 32 output tiles, the largest eligible
    dict(B=32, T=70, F=33, r=100, K=3, ah=True), # two row tiles x 13 output tiles
    dict(B=20, T=9, F=65, r=128, K=4),           # two row tiles x 16 output tiles = 32 workgroups
    dict(B=32, T=130, F=65, r=250, K=3),         # two chains x 32 output tiles, > 2 blocks of frames
    dict(B=128, T=66, F=33, r=24, K=5, ah=True), # eight chains (one per XCD) x 2 tiles, all-hidden
    dict(B=100, T=10, F=21, r=100, K=2),         # seven chains, the last with 4 live rows
    dict(B=250, T=70, F=33, r=100, K=3),         # the reference's inference slab: 16 chains, two per XCD
    dict(B=200, T=9, F=21, r=120, K=4, ah=True), # 13 chains x 15 tiles
])
def test_persistent_gram_kernel_is_bit_identical(dev, monkeypatch, cfg):
    """gram_persist_kernel (cell_gram_persist.h): one launch per block of frames, every 16-row tile an
    independent chain of <= 32 workgroups on its own XCD with an in-kernel barrier between
    layer-steps -- the same arithmetic in the same order as the
    launch-per-layer-step Gram kernels, so the two must agree bit for bit (ragged lengths, a masked
    first frame, all-hidden output, stateful continuation); both are checked against the fp64
    oracle by the parity suites."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    B, T, F, r, K = cfg["B"], cfg["T"], cfg["F"], cfg["r"], cfg["K"]
    ah = cfg.get("ah", False)
    P, alt, labels, N = TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), ragged=B > 1,
                                    seed=B + T)
    if B > 1:
        P["X"][B - 1, 0] = -1.0                   # a masked FIRST frame
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DRNMF_GRAM", "1")
        monkeypatch.setenv("DRNMF_PERSIST", mode)
        h, _, _ = TP._run_cell(dev, P, alt, labels, N, K, return_all_hidden=ah)
        outs[mode] = h.copy()
    assert np.array_equal(outs["0"], outs["1"])
    ref = TP._oracle_cell(P, alt, labels, K, return_all_hidden=ah)
    assert np.max(np.abs(outs["1"] - ref)) <= TP.H_TOL * np.max(np.abs(ref))


@pytest.mark.parametrize("cfg", [
    dict(B=3, T=150, F=21, r=6, K=2),             # one chain, K = 2: edge + one layer-step per frame
    dict(B=7, T=70, F=33, r=50, K=3),             # ragged, 7 output tiles
    dict(B=32, T=131, F=65, r=100, K=5),          # the shipped r = 100 batch: two chains x 13 tiles
    dict(B=128, T=20, F=33, r=24, K=4),           # eight chains
    dict(B=5, T=9, F=21, r=250, K=3),             # N = 500: 32 tiles per chain
    dict(B=250, T=12, F=33, r=100, K=3),          # 16 chains, two per XCD
])
def test_persistent_gram_bptt_is_bit_identical(dev, monkeypatch, cfg):
    """gram_persist_bwd_kernel (cell_gram_persist.h): the BPTT's whole sequential pass in one launch of
    independent per-row-tile chains, against bwd_edge_kernel + gram_bwd_kernel replayed from
    hipGraphs -- the same arithmetic in the same order: every gradient and the loss bit for bit (the
    graph form is checked against fp64 autograd of the oracle in tests/test_gpu_train.py, T up to
    322)."""
    from drnmf_amd import layers
    B, T, F, r, K = cfg["B"], cfg["T"], cfg["F"], cfg["r"], cfg["K"]
    P = O.synth_problem(B, T, F, r, seed=B + T, ragged=True, density=0.15)
    P["X"][B - 1, 0] = -1.0                       # a masked first frame
    P["Y"][B - 1, 0] = -1.0
    N = 2 * r
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=P["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
             params_trainable=["log_D", "log_alph", "log_lam1"])
    w = (P["X"] != -1.0).any(-1).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DRNMF_GRAM", "1")
        monkeypatch.setenv("DRNMF_PERSIST", mode)
        np.random.seed(5)
        model = layers.build_unfolded_snmf(p, device=dev)
        model.compile(lr=1e-3)
        res[mode] = model.loss_and_grads(t(P["X"]), t(P["Y"]), t(w)).cpu().numpy().copy()
    assert np.isfinite(res["1"]).all() and np.abs(res["1"][:-4]).max() > 0
    assert np.array_equal(res["0"], res["1"])


def test_persistent_chains_from_concurrent_streams(dev, monkeypatch):
    """persistent launches issued from several streams at once.  The handle admits one
    stream's persistent launches at a time (the others take the launch-per-layer-step graphs, which
    compute the same bits), so four concurrent streams must all finish, agree with a serial run, and
    leave no timeout flag behind."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    monkeypatch.setenv("DRNMF_GRAM", "1")
    monkeypatch.setenv("DRNMF_PERSIST", "1")
    B, T, F, r, K = 16, 192, 33, 100, 4
    probs = [TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), ragged=True, seed=40 + i)
             for i in range(4)]
    from drnmf_amd import ops
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    serial, prepared = [], []
    for P, alt, labels, N in probs:
        h, params, desc = TP._run_cell(dev, P, alt, labels, N, K)
        serial.append(h.copy())
        prepared.append((t(P["X"]), params, desc, t(P["log_h0"]), O.u_scalars(alt, np.float32)))
    # one host thread (a handle is not re-entrant), four streams: the launches of the four calls
    # overlap on the GPU
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    outs = [None] * 4
    torch.cuda.synchronize()
    for _ in range(3):
        for i, (x, params, desc, lh0, u) in enumerate(prepared):
            with torch.cuda.stream(streams[i]):
                outs[i] = ops.cell_forward(x, -1.0, params, desc, lh0, u)
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(outs[i].cpu().numpy(), serial[i]), i
    # and the handle still works (no timeout was recorded)
    again = TP._run_cell(dev, *probs[0], K)[0]
    assert np.array_equal(again, serial[0])


def test_fuzz_persistent_chains_bit_identical(dev, monkeypatch):
    """Seeded sweep of the shapes the persistent chains accept (1-8 row tiles, 1-32 output tiles, K = 2..6,
    T on both sides of the 64-frame blocks and of the all-frames-resident switch, ragged, all-hidden):
    forward outputs and every gradient bit for bit against the launch-per-layer-step graphs."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    from drnmf_amd import layers
    rng = np.random.default_rng(303)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    for it in range(14):
        B = int(rng.choice([1, 5, 16, 17, 40, 100, 128]))
        T = int(rng.choice([1, 2, 63, 64, 65, 129, 200]))
        F = int(rng.choice([17, 33, 65]))
        r = int(rng.choice([4, 8, 20, 60, 100, 250]))
        K = int(rng.integers(2, 7))
        if B * T * r > 600000:
            T = max(1, 600000 // (B * r))
        ah = bool(rng.integers(0, 2))
        P, alt, labels, N = TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), ragged=B > 1,
                                        seed=100 + it)
        outs = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("DRNMF_GRAM", "1")
            monkeypatch.setenv("DRNMF_PERSIST", mode)
            outs[mode] = TP._run_cell(dev, P, alt, labels, N, K, return_all_hidden=ah)[0].copy()
        assert np.array_equal(outs["0"], outs["1"]), dict(B=B, T=T, F=F, r=r, K=K, ah=ah, it=it)
        if it % 2:
            continue
        # gradients through the model surface
        Pm = O.synth_problem(B, T, F, r, seed=200 + it, ragged=True, density=0.15)
        p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
                 W=Pm["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
                 params_trainable=["log_D", "log_alph", "log_lam1"])
        w = (Pm["X"] != -1.0).any(-1).astype(np.float32)
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("DRNMF_PERSIST", mode)
            np.random.seed(5)
            model = layers.build_unfolded_snmf(p, device=dev)
            model.compile(lr=1e-3)
            res[mode] = model.loss_and_grads(t(Pm["X"]), t(Pm["Y"]), t(w)).cpu().numpy().copy()
        assert np.array_equal(res["0"], res["1"]), dict(B=B, T=T, F=F, r=r, K=K, it=it)


def test_persistent_chains_next_to_a_busy_stream(dev, monkeypatch):
    """the chains' in-kernel barriers with ANOTHER stream occupying the CUs (here a
    stream of large GEMMs, as an RCCL kernel or a second model would): the persistent launches must
    complete -- late if their workgroups have to wait for CUs, never wrong -- and leave no timeout."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    from drnmf_amd import ops
    monkeypatch.setenv("DRNMF_GRAM", "1")
    monkeypatch.setenv("DRNMF_PERSIST", "1")
    B, T, F, r, K = 32, 256, 33, 100, 4
    P, alt, labels, N = TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), ragged=True, seed=77)
    ref, params, desc = TP._run_cell(dev, P, alt, labels, N, K)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    x, lh0, u = t(P["X"]), t(P["log_h0"]), O.u_scalars(alt, np.float32)
    a = torch.randn((8192, 8192), device=dev)
    b = torch.randn((8192, 8192), device=dev)
    torch.cuda.synchronize()
    busy, work = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = []
    for rep in range(3):
        with torch.cuda.stream(busy):
            for _ in range(6):
                c = a @ b                                   # ~1.1 TFLOP each: the chip is full for milliseconds
        with torch.cuda.stream(work):
            outs.append(ops.cell_forward(x, -1.0, params, desc, lh0, u))
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), ref)
    again = TP._run_cell(dev, P, alt, labels, N, K)[0]
    from drnmf_amd import ops
    ops.check_status(dev)                                   # no DRNMF_ERR_TIMEOUT pending
    assert np.array_equal(again, ref)


def test_persistent_timeout_is_reported_not_swallowed(dev, monkeypatch):
    """a persistent launch whose barrier can never complete (fault injection: every barrier
    waits for one arrival too many) must give up after its bounded spin, and drnmf_check_status -- called
    once the stream is synchronised -- must fail with DRNMF_ERR_TIMEOUT instead of a wrong result going out
    silently (round 4: no longer reported implicitly by the next cell call, which raced with the training
    step's device-side guard); the call after that works again."""
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    from drnmf_amd import _capi
    monkeypatch.setenv("DRNMF_GRAM", "1")
    monkeypatch.setenv("DRNMF_PERSIST", "1")
    from drnmf_amd import ops as _ops
    if not _ops.persist_admitted(dev):       # (the fault is injected into the persistent chains only)
        pytest.skip("not admitted to the persistent chains: " + _ops.persist_admit_reason(dev))
    B, T, F, r, K = 3, 4, 21, 6, 2
    P, alt, labels, N = TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), seed=9)
    good = TP._run_cell(dev, P, alt, labels, N, K)[0].copy()
    monkeypatch.setenv("DRNMF_PERSIST_FAULT", "1")
    t0 = time.time()
    TP._run_cell(dev, P, alt, labels, N, K)               # enqueues; the kernel times out on the device
    torch.cuda.synchronize()
    assert time.time() - t0 < 60.0                         # bounded, not a hang
    monkeypatch.delenv("DRNMF_PERSIST_FAULT")
    from drnmf_amd import ops
    with pytest.raises(_capi.DrnmfError, match="timed out"):
        ops.check_status(dev)                              # (after the synchronise above: read and clear)
    ops.check_status(dev)                                  # the fault is reported once
    again = TP._run_cell(dev, P, alt, labels, N, K)[0]
    ops.check_status(dev)
    assert np.array_equal(again, good)


def _tenant_worker(rank, world, out_dir):
    """One of several PROCESSES sharing the GPU, each running a chain-eligible forward over and over."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    from drnmf_amd import ops
    torch.cuda.set_device(0)
    _session_matrix_mode()
    os.environ["DRNMF_GRAM"] = "1"
    os.environ["DRNMF_PERSIST"] = "1"
    ops.reload_env()
    dev = torch.device("cuda:0")
    B, T, F, r, K = 16, 192, 33, 100, 4
    P, alt, labels, N = TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), ragged=True, seed=40)
    admitted = ops.persist_admitted(dev)
    outs = []
    for _ in range(12):
        h, _, _ = TP._run_cell(dev, P, alt, labels, N, K)
        outs.append(h)
    ops.check_status(dev)                       # raises on a chain that timed out
    for h in outs[1:]:
        assert np.array_equal(h, outs[0])
    np.savez(os.path.join(out_dir, "tenant%d.npz" % rank), h=outs[0], admitted=np.array([int(admitted)]))


def test_processes_sharing_one_gpu_admit_one_owner_of_the_persistent_chains(dev, monkeypatch, tmp_path):
    """persistent chains of two PROCESSES on one GPU are not coordinated on
    the device (each could hold CUs the other's workgroups need).  Policy: the first handle on a device
    holds an exclusive flock for its lifetime and is the only one admitted to the chains
    (drnmf_persist_admitted); every other process runs the launch-per-layer-step graphs -- the same bits.
    Here this pytest process is the owner; three concurrent tenant processes must not be admitted, must
    finish without a timeout and must reproduce the owner's output bit for bit."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as TP
    from drnmf_amd import ops
    monkeypatch.setenv("DRNMF_GRAM", "1")
    monkeypatch.setenv("DRNMF_PERSIST", "1")
    if not ops.persist_admitted(dev):        # (a shared box: another process or user owns the device's lock file)
        pytest.skip("this process was not admitted to the persistent chains: " + ops.persist_admit_reason(dev))
    B, T, F, r, K = 16, 192, 33, 100, 4
    P, alt, labels, N = TP._problem(B, T, F, r, K, untied=("log_D", "log_alph"), ragged=True, seed=40)
    mine, _, _ = TP._run_cell(dev, P, alt, labels, N, K)
    ops.check_status(dev)
    mp.spawn(_tenant_worker, args=(3, str(tmp_path)), nprocs=3, join=True)
    for rank in range(3):
        z = np.load(tmp_path / ("tenant%d.npz" % rank))
        assert int(z["admitted"][0]) == 0
        assert np.array_equal(z["h"], mine)
    ref = TP._oracle_cell(P, alt, labels, K)
    assert np.max(np.abs(mine - ref)) <= TP.H_TOL * np.max(np.abs(ref))


def test_timed_out_training_step_is_skipped_and_reported(dev, monkeypatch):
    """a persistent chain that times out inside a TRAINING step must not feed garbage
    gradients into Adam.  Fault injection (DRNMF_PERSIST_FAULT=1: every chain barrier waits for an arrival
    that never comes): the step's fault word -- taken from the handle stream-ordered, part of the
    all-reduced flat buffer -- makes the fused Adam launch skip the update (weights and moments
    bit-identical to before the step), float(loss) raises, check_status finds nothing left behind, and the
    next step trains normally; predict_on_batch / test_on_batch report a timed-out forward themselves."""
    from drnmf_amd import _capi, layers, ops
    monkeypatch.setenv("DRNMF_GRAM", "1")
    monkeypatch.setenv("DRNMF_PERSIST", "1")
    if not ops.persist_admitted(dev):        # (the fault is injected into the persistent chains only)
        pytest.skip("not admitted to the persistent chains: " + ops.persist_admit_reason(dev))
    B, T, F, r, K = 5, 6, 21, 6, 3
    P = O.synth_problem(B, T, F, r, seed=31, ragged=True, density=0.15)
    N = 2 * r
    p = dict(input_dim=F, hidden_dim=N, output_dim=F, mask_value=-1., maxseq=T, K_layers=K,
             W=P["W"], alph=N / 4.0, lam1=0.3, params_untied=["log_D", "log_alph"],
             params_trainable=["log_D", "log_alph"])
    np.random.seed(3)
    model = layers.build_unfolded_snmf(p, device=dev)
    model.compile(lr=1e-2)
    w = (P["X"] != -1.0).any(-1).astype(np.float32)
    l0 = float(model.train_on_batch(P["X"], P["Y"], w))
    assert np.isfinite(l0)
    before = [a.copy() for a in model.get_weights()]
    m_before = model._mflat.clone()
    it_before = model.opt["iterations"]
    monkeypatch.setenv("DRNMF_PERSIST_FAULT", "1")
    bad = model.train_on_batch(P["X"], P["Y"], w)          # enqueues; the chains time out on the device
    monkeypatch.delenv("DRNMF_PERSIST_FAULT")
    with pytest.raises(_capi.DrnmfError, match="timed out"):
        float(bad)
    torch.cuda.synchronize()
    for a, b in zip(before, model.get_weights()):
        assert np.array_equal(a, b)                          # the update was skipped
    assert torch.equal(m_before, model._mflat)
    assert model.opt["iterations"] == it_before              # (a skipped step is not an iteration: ADVICE r4)
    assert float(model._step_dev[model._step_no & 1]) == it_before      # ... on the device, where lr_t is made
    ops.check_status(dev)                                    # the fault word was consumed by the step
    l2 = float(model.train_on_batch(P["X"], P["Y"], w))
    assert np.isfinite(l2) and any(not np.array_equal(a, b) for a, b in zip(before, model.get_weights()))
    assert model.opt["iterations"] == it_before + 1
    # a fault left behind by an ASYNCHRONOUS inference call nobody checked is not charged to the next
    # training / test step: the step takes the stale word into a scratch first
    xt = torch.from_numpy(P["X"]).to(dev)
    monkeypatch.setenv("DRNMF_PERSIST_FAULT", "1")
    model.forward(xt)                                        # times out on the device; no status check
    monkeypatch.delenv("DRNMF_PERSIST_FAULT")
    l3 = float(model.train_on_batch(P["X"], P["Y"], w))      # healthy step: neither skipped nor raised
    assert np.isfinite(l3) and model.opt["iterations"] == it_before + 2
    assert float(model._stale_faults.item()) == 1.0
    monkeypatch.setenv("DRNMF_PERSIST_FAULT", "1")
    model.forward(xt)
    monkeypatch.delenv("DRNMF_PERSIST_FAULT")
    assert np.isfinite(model.test_on_batch(P["X"], P["Y"], w))
    ops.check_status(dev)
    # inference: the copy to the host synchronises, then the status check raises
    monkeypatch.setenv("DRNMF_PERSIST_FAULT", "1")
    with pytest.raises(_capi.DrnmfError, match="timed out"):
        model.predict_on_batch(P["X"])
    with pytest.raises(_capi.DrnmfError, match="timed out"):
        model.test_on_batch(P["X"], P["Y"], w)
    monkeypatch.delenv("DRNMF_PERSIST_FAULT")
    assert np.isfinite(model.predict_on_batch(P["X"])).all()
