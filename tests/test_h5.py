"""Keras HDF5 weight files (enhance.py:1096, 1119-1129, 1135, 1160-1166) through drnmf_amd.h5lite,
the ctypes binding of the system's libhdf5 used when h5py is not importable.  Files are checked
with the HDF5 command-line tools (h5dump) when the image has them."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import drnmf_oracle as O


@pytest.fixture(scope="module")
def h5():
    from drnmf_amd import h5lite
    if not h5lite.available():
        pytest.skip("no libhdf5 >= 1.10 on this machine")
    return h5lite


def _h5dump():
    for cand in (shutil.which("h5dump"), "/opt/conda/bin/h5dump"):
        if cand and os.path.exists(cand):
            return cand
    return None


def test_h5lite_round_trip_and_h5dump(h5, tmp_path):
    path = str(tmp_path / "t.h5")
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3, 5)).astype(np.float32)
    b = rng.standard_normal((4,))                       # float64
    with h5.File(path, "w") as f:
        f.attrs["layer_names"] = [b"first_layer", b"b"]
        f.attrs["backend"] = b"theano"
        f.attrs["n"] = np.array([3, 4])
        g = f.create_group("first_layer")
        g.attrs["weight_names"] = ["first_layer_W", "bias"]      # str is accepted too
        g.create_dataset("first_layer_W", data=a)
        g.create_dataset("bias", data=b)
        f.create_group("b").attrs["weight_names"] = np.array([], dtype="S1")
    with h5.File(path, "r") as f:
        assert list(f.attrs["layer_names"]) == [b"first_layer", b"b"]
        assert f.attrs["backend"] == b"theano"
        np.testing.assert_array_equal(f.attrs["n"], [3, 4])
        assert "first_layer" in f and "nope" not in f and "nope" not in f.attrs
        g = f["first_layer"]
        assert list(g.attrs["weight_names"]) == [b"first_layer_W", b"bias"]
        w = np.asarray(g["first_layer_W"])
        assert w.dtype == np.float32 and g["first_layer_W"].shape == (3, 5)
        np.testing.assert_array_equal(w, a)
        np.testing.assert_array_equal(np.asarray(g["bias"]), b)
        assert len(f["b"].attrs["weight_names"]) == 0
        with pytest.raises(KeyError):
            f["nope"]
    with pytest.raises(IOError):
        h5.File(str(tmp_path / "missing.h5"), "r")
    tool = _h5dump()
    if tool:
        txt = subprocess.run([tool, path], capture_output=True, text=True, check=True).stdout
        assert 'ATTRIBUTE "layer_names"' in txt and '"first_layer", "b' in txt
        assert "H5T_STR_NULLPAD" in txt and 'DATASET "first_layer_W"' in txt
        assert "H5T_IEEE_F32LE" in txt and "H5T_IEEE_F64LE" in txt


def test_variable_length_string_attributes_are_read(h5, tmp_path):
    """Newer h5py/Keras write str attributes as variable-length strings."""
    path = str(tmp_path / "v.h5")
    L = h5.lib()
    with h5.File(path, "w") as f:
        tid = L.H5Tcopy(h5._T["c_s1"])
        L.H5Tset_size(tid, h5.H5T_VARIABLE)
        dims = (C.c_uint64 * 1)(2)
        sid = L.H5Screate_simple(1, dims, None)
        aid = L.H5Acreate2(f._id, b"layer_names", tid, sid, 0, 0)
        buf = (C.c_char_p * 2)(b"alpha", b"be")
        assert L.H5Awrite(aid, tid, buf) >= 0
        L.H5Aclose(aid); L.H5Sclose(sid); L.H5Tclose(tid)
    with h5.File(path, "r") as f:
        assert list(f.attrs["layer_names"]) == [b"alpha", b"be"]


def _model():
    from drnmf_amd import layers
    P = O.synth_problem(2, 3, 21, 6, seed=5)
    p = dict(input_dim=21, hidden_dim=12, output_dim=21, mask_value=-1., maxseq=3, K_layers=2,
             W=P["W"], alph=3.0, lam1=0.3, params_untied=["log_D"], params_trainable=["log_D"])
    return layers.build_unfolded_snmf(p, device="cpu")


def test_model_weights_hdf5_round_trip(h5, tmp_path):
    m = _model()
    path = str(tmp_path / "weights.hdf5")
    m.save_weights(path)
    w0 = m.get_weights()
    m.set_weights([a + 1 for a in w0])
    m.load_weights(path)
    for a, b in zip(w0, m.get_weights()):
        np.testing.assert_array_equal(a, b)
        assert a.shape == b.shape
    # build_alt's scalar parameters (log_alph, log_lam1: np.log of a float32, enhance.py:147) are 0-d arrays;
    # K.variable keeps that shape and h5py writes a scalar dataspace: the layer, the tree and the file agree
    names = [str(n) for n in m.weights_tree()[m.cell.name + "/weight_names"]]
    scalars = [w for n, w in zip(names, m.cell.get_weights()) if n.endswith(("log_alph", "log_lam1"))]
    assert len(scalars) == 2 and all(w.shape == () for w in scalars)
    with h5.File(path, "r") as f:
        for n in names:
            if n.endswith(("log_alph", "log_lam1")):
                assert tuple(f[m.cell.name][n].shape) == ()
    tool = _h5dump()
    if tool:
        txt = subprocess.run([tool, "-A", path], capture_output=True, text=True, check=True).stdout
        assert '"2.0.4"' in txt and '"theano"' in txt
        assert 'GROUP "%s"' % m.cell.name in txt and 'GROUP "clean_est"' in txt
        assert 'DATASET "%s_log_h0"' % m.cell.name in txt and 'DATASET "kernel"' in txt


def test_reference_style_file_is_loaded(h5, tmp_path):
    """A file as the reference's ModelCheckpoint / model.save would leave it: the reference's
    layer names, the cell's weights in another order, ':0' suffixes tolerated, and the
    `model_weights` subgroup of a full-model file."""
    m = _model()
    tree = m.weights_tree()
    cn = m.cell.name
    wn = [str(n) for n in tree[cn + "/weight_names"]]
    path = str(tmp_path / "model.h5")
    with h5.File(path, "w") as f:
        g0 = f.create_group("model_weights")
        g0.attrs["layer_names"] = [b"masking_1", b"simple_deep_rnn_7", b"time_distributed_1",
                                   b"time_distributed_2"]
        g0.create_group("masking_1").attrs["weight_names"] = np.array([], dtype="S1")
        g = g0.create_group("simple_deep_rnn_7")
        names = ["simple_deep_rnn_7" + n[len(cn):] for n in reversed(wn)]
        g.attrs["weight_names"] = [n.encode() for n in names]
        for n, src in zip(names, reversed(wn)):
            v = tree[cn + "/" + src] * 3
            # (a scalar weight written as (1,) by some other tool is the same scalar)
            g.create_dataset(n, data=v.reshape(1) if v.ndim == 0 and n.endswith("log_lam1") else v)
        for dst, src in (("time_distributed_1", "clean_est"), ("time_distributed_2", "noise_est")):
            gg = g0.create_group(dst)
            gg.attrs["weight_names"] = [b"kernel"]
            gg.create_dataset("kernel", data=tree[src + "/kernel"] * 3)
    w0 = m.get_weights()
    m.load_weights(path)
    for a, b in zip(w0, m.get_weights()):
        np.testing.assert_array_equal(3 * a, b)
