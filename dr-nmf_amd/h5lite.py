"""Minimal HDF5 access for Keras weight files, bound with ctypes to the system's libhdf5 (the C
library h5py itself wraps; HDF5 1.10 / 1.12 API).  Used when h5py is not importable; exposes the
few h5py idioms `UnfoldedSNMFModel.save_weights / load_weights` need:

    with File(path, 'w') as f:
        f.attrs['layer_names'] = [b'a', b'b']        # fixed-length string array, as h5py stores it
        g = f.create_group('a'); g.attrs['weight_names'] = [...]
        g.create_dataset('kernel', data=ndarray)
    with File(path, 'r') as f:
        f.attrs['layer_names'];  'model_weights' in f;  np.asarray(f['a']['kernel'])

Reference: the Keras 2.0.4 HDF5 weight files of enhance.py:1096, 1119-1129, 1135, 1160-1166
(ModelCheckpoint(save_weights_only=True), save_weights, load_weights).  Host-side file format code:
no device work happens here.
"""
import ctypes as C
import ctypes.util
import glob
import os
import weakref

import numpy as np

_hid = C.c_int64
_lib = None
_T = {}

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_VARIABLE = C.c_size_t(-1).value


class H5Error(IOError):
    pass


def _candidates():
    env = os.environ.get("DRNMF_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*",
                "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def available():
    try:
        lib()
        return True
    except ImportError:
        return False


def lib():
    """Load libhdf5 once (ImportError if the system has none)."""
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for cand in _candidates():
        try:
            L = C.CDLL(cand)
            L.H5open
        except (OSError, AttributeError) as e:
            last = e
            continue
        sig = {
            "H5open": (C.c_int, []),
            "H5get_libversion": (C.c_int, [C.POINTER(C.c_uint)] * 3),
            "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
            "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
            "H5Fclose": (C.c_int, [_hid]),
            "H5Gcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid]),
            "H5Gopen2": (_hid, [_hid, C.c_char_p, _hid]),
            "H5Gclose": (C.c_int, [_hid]),
            "H5Oopen": (_hid, [_hid, C.c_char_p, _hid]),
            "H5Oclose": (C.c_int, [_hid]),
            "H5Iget_type": (C.c_int, [_hid]),
            "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
            "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
            "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
            "H5Dget_space": (_hid, [_hid]),
            "H5Dget_type": (_hid, [_hid]),
            "H5Dclose": (C.c_int, [_hid]),
            "H5Dvlen_reclaim": (C.c_int, [_hid, _hid, _hid, C.c_void_p]),
            "H5Screate_simple": (_hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "H5Screate": (_hid, [C.c_int]),
            "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(C.c_uint64),
                                                     C.POINTER(C.c_uint64)]),
            "H5Sclose": (C.c_int, [_hid]),
            "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]),
            "H5Aopen": (_hid, [_hid, C.c_char_p, _hid]),
            "H5Aexists": (C.c_int, [_hid, C.c_char_p]),
            "H5Awrite": (C.c_int, [_hid, _hid, C.c_void_p]),
            "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]),
            "H5Aget_space": (_hid, [_hid]),
            "H5Aget_type": (_hid, [_hid]),
            "H5Aclose": (C.c_int, [_hid]),
            "H5Adelete": (C.c_int, [_hid, C.c_char_p]),
            "H5Tcopy": (_hid, [_hid]),
            "H5Tset_size": (C.c_int, [_hid, C.c_size_t]),
            "H5Tset_strpad": (C.c_int, [_hid, C.c_int]),
            "H5Tget_size": (C.c_size_t, [_hid]),
            "H5Tget_class": (C.c_int, [_hid]),
            "H5Tis_variable_str": (C.c_int, [_hid]),
            "H5Tclose": (C.c_int, [_hid]),
            "H5Eset_auto2": (C.c_int, [_hid, C.c_void_p, C.c_void_p]),
        }
        try:
            for name, (res, args) in sig.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
        except AttributeError as e:
            last = e
            continue
        if L.H5open() < 0:
            last = OSError("H5open failed in %s" % cand)
            continue
        mj, mn, rl = C.c_uint(), C.c_uint(), C.c_uint()
        L.H5get_libversion(C.byref(mj), C.byref(mn), C.byref(rl))
        if (mj.value, mn.value) < (1, 10):           # hid_t is 32 bits before 1.10
            last = OSError("%s is HDF5 %d.%d; 1.10 or newer is needed" % (cand, mj.value, mn.value))
            continue
        L.H5Eset_auto2(0, None, None)                # errors are reported by return codes here
        for key, sym in (("f32", "H5T_NATIVE_FLOAT_g"), ("f64", "H5T_NATIVE_DOUBLE_g"),
                         ("i32", "H5T_NATIVE_INT32_g"), ("i64", "H5T_NATIVE_INT64_g"),
                         ("c_s1", "H5T_C_S1_g")):
            _T[key] = _hid.in_dll(L, sym).value
        _lib = L
        return _lib
    raise ImportError("no usable libhdf5 (>= 1.10) found for Keras HDF5 weight files (tried "
                      "$DRNMF_HDF5_LIB, the loader path, /usr/lib*, /opt/conda/lib)%s"
                      % ("; last error: %s" % last if last else ""))


def _chk(v, what):
    if v < 0:
        raise H5Error("HDF5: %s failed" % what)
    return v


def _space_shape(L, sid):
    nd = _chk(L.H5Sget_simple_extent_ndims(sid), "H5Sget_simple_extent_ndims")
    dims = (C.c_uint64 * max(nd, 1))()
    if nd:
        _chk(L.H5Sget_simple_extent_dims(sid, dims, None), "H5Sget_simple_extent_dims")
    return tuple(int(dims[i]) for i in range(nd))


def _np_mem_type(L, tid):
    """(numpy dtype, HDF5 memory type id) to read a numeric file type with."""
    cls, size = L.H5Tget_class(tid), L.H5Tget_size(tid)
    if cls == H5T_FLOAT:
        return (np.float32, _T["f32"]) if size <= 4 else (np.float64, _T["f64"])
    if cls == H5T_INTEGER:
        return (np.int32, _T["i32"]) if size <= 4 else (np.int64, _T["i64"])
    raise H5Error("HDF5: unsupported datatype class %d" % cls)


def _read_strings(L, read, tid, sid):
    """String attribute/dataset -> bytes (scalar) or ndarray of bytes objects."""
    shape = _space_shape(L, sid)
    n = int(np.prod(shape)) if shape else 1
    if L.H5Tis_variable_str(tid) > 0:
        buf = (C.c_char_p * n)()
        mt = L.H5Tcopy(_T["c_s1"])
        L.H5Tset_size(mt, H5T_VARIABLE)
        _chk(read(mt, buf), "read (variable-length strings)")
        vals = [bytes(buf[i]) if buf[i] is not None else b"" for i in range(n)]
        L.H5Dvlen_reclaim(mt, sid, 0, buf)
        L.H5Tclose(mt)
    else:
        size = L.H5Tget_size(tid)
        raw = C.create_string_buffer(size * n)
        _chk(read(tid, raw), "read (fixed-length strings)")
        vals = [raw.raw[i * size:(i + 1) * size].split(b"\x00", 1)[0].rstrip(b" ")
                for i in range(n)]
    if not shape:
        return vals[0]
    out = np.empty(n, dtype=object)
    out[:] = vals
    return out.reshape(shape)


class _Attrs(object):
    def __init__(self, obj):
        self._o = obj

    def __contains__(self, name):
        return lib().H5Aexists(self._o._id, name.encode()) > 0

    def __getitem__(self, name):
        L = lib()
        if name not in self:
            raise KeyError(name)
        aid = _chk(L.H5Aopen(self._o._id, name.encode(), 0), "H5Aopen(%s)" % name)
        tid, sid = L.H5Aget_type(aid), L.H5Aget_space(aid)
        try:
            if L.H5Tget_class(tid) == H5T_STRING:
                return _read_strings(L, lambda mt, buf: L.H5Aread(aid, mt, buf), tid, sid)
            dt, mt = _np_mem_type(L, tid)
            out = np.empty(_space_shape(L, sid), dtype=dt)
            _chk(L.H5Aread(aid, mt, out.ctypes.data_as(C.c_void_p)), "H5Aread(%s)" % name)
            return out if out.shape else out[()]
        finally:
            L.H5Tclose(tid)
            L.H5Sclose(sid)
            L.H5Aclose(aid)

    def __setitem__(self, name, value):
        L = lib()
        if name in self:
            L.H5Adelete(self._o._id, name.encode())
        if isinstance(value, str):
            value = value.encode("utf8")
        if isinstance(value, bytes):
            items, shape = [value], ()
        else:
            arr = np.asarray(value)
            if arr.dtype.kind in "SUO":
                items = [v.encode("utf8") if isinstance(v, str) else bytes(v)
                         for v in arr.reshape(-1).tolist()]
                shape = arr.shape
            else:
                items = None
        if items is not None:                              # fixed-length strings, null padded
            size = max([len(v) for v in items] + [1])
            tid = L.H5Tcopy(_T["c_s1"])
            L.H5Tset_size(tid, size)
            L.H5Tset_strpad(tid, 1)                        # H5T_STR_NULLPAD, as h5py's numpy 'S'
            buf = C.create_string_buffer(b"".join(v.ljust(size, b"\x00") for v in items),
                                         size * max(len(items), 1))
            ptr = C.cast(buf, C.c_void_p)
        else:
            arr = np.ascontiguousarray(arr, dtype={"f": np.float32 if arr.dtype.itemsize <= 4
                                                   else np.float64}.get(arr.dtype.kind, np.int64)
                                       ).reshape(arr.shape)      # (ascontiguousarray makes 0-d arrays (1,))
            tid = L.H5Tcopy(_T[{np.dtype(np.float32): "f32", np.dtype(np.float64): "f64",
                                np.dtype(np.int64): "i64"}[arr.dtype]])
            shape = arr.shape
            ptr = arr.ctypes.data_as(C.c_void_p)
        if shape:
            dims = (C.c_uint64 * len(shape))(*shape)
            sid = L.H5Screate_simple(len(shape), dims, None)
        else:
            sid = L.H5Screate(H5S_SCALAR)
        aid = _chk(L.H5Acreate2(self._o._id, name.encode(), tid, sid, 0, 0), "H5Acreate2(%s)" % name)
        try:
            _chk(L.H5Awrite(aid, tid, ptr), "H5Awrite(%s)" % name)
        finally:
            L.H5Aclose(aid)
            L.H5Sclose(sid)
            L.H5Tclose(tid)


class Dataset(object):
    def __init__(self, did, file=None):
        self._id = did
        self.attrs = _Attrs(self)
        if file is not None:
            file._children.add(self)      # closed with the file (an open object keeps it locked)

    def _read(self):
        L = lib()
        tid, sid = L.H5Dget_type(self._id), L.H5Dget_space(self._id)
        try:
            if L.H5Tget_class(tid) == H5T_STRING:
                return _read_strings(L, lambda mt, buf: L.H5Dread(self._id, mt, 0, 0, 0, buf), tid,
                                     sid)
            dt, mt = _np_mem_type(L, tid)
            out = np.empty(_space_shape(L, sid), dtype=dt)
            _chk(L.H5Dread(self._id, mt, 0, 0, 0, out.ctypes.data_as(C.c_void_p)), "H5Dread")
            return out
        finally:
            L.H5Tclose(tid)
            L.H5Sclose(sid)

    @property
    def shape(self):
        L = lib()
        sid = L.H5Dget_space(self._id)
        try:
            return _space_shape(L, sid)
        finally:
            L.H5Sclose(sid)

    def __array__(self, dtype=None, copy=None):
        a = self._read()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, key):
        return self._read()[key]

    def close(self):
        if self._id is not None:
            lib().H5Dclose(self._id)
            self._id = None

    __del__ = close


class Group(object):
    def __init__(self, gid, owner=True, file=None):
        self._id = gid
        self._owner = owner
        self._file = file
        self.attrs = _Attrs(self)
        if file is not None:
            file._children.add(self)

    def __contains__(self, name):
        return lib().H5Lexists(self._id, name.encode(), 0) > 0

    def create_group(self, name):
        return Group(_chk(lib().H5Gcreate2(self._id, name.encode(), 0, 0, 0),
                          "H5Gcreate2(%s)" % name), file=self._file)

    def create_dataset(self, name, data):
        L = lib()
        arr = np.asarray(data)
        # (0-d arrays -- Keras' scalar weights, build_alt's log_alph / log_lam1 -- keep a scalar dataspace as
        # h5py writes them: np.ascontiguousarray alone returns shape (1,))
        if arr.dtype.kind == "f":
            arr = np.ascontiguousarray(arr, np.float32 if arr.dtype.itemsize <= 4 else np.float64).reshape(arr.shape)
            key = "f32" if arr.dtype == np.float32 else "f64"
        elif arr.dtype.kind in "iub":
            arr = np.ascontiguousarray(arr, np.int64).reshape(arr.shape)
            key = "i64"
        else:
            raise TypeError("create_dataset: unsupported dtype %s" % arr.dtype)
        if arr.shape:
            dims = (C.c_uint64 * arr.ndim)(*arr.shape)
            sid = L.H5Screate_simple(arr.ndim, dims, None)
        else:
            sid = L.H5Screate(H5S_SCALAR)
        did = _chk(L.H5Dcreate2(self._id, name.encode(), _T[key], sid, 0, 0, 0),
                   "H5Dcreate2(%s)" % name)
        try:
            _chk(L.H5Dwrite(did, _T[key], 0, 0, 0, arr.ctypes.data_as(C.c_void_p)),
                 "H5Dwrite(%s)" % name)
        finally:
            L.H5Sclose(sid)
        return Dataset(did, file=self._file)

    def __getitem__(self, name):
        L = lib()
        if name not in self:
            raise KeyError(name)
        oid = _chk(L.H5Oopen(self._id, name.encode(), 0), "H5Oopen(%s)" % name)
        kind = L.H5Iget_type(oid)              # H5I_GROUP = 2, H5I_DATASET = 5
        L.H5Oclose(oid)
        if kind == 2:
            return Group(_chk(L.H5Gopen2(self._id, name.encode(), 0), "H5Gopen2(%s)" % name),
                         file=self._file)
        if kind == 5:
            return Dataset(_chk(L.H5Dopen2(self._id, name.encode(), 0), "H5Dopen2(%s)" % name),
                           file=self._file)
        raise H5Error("HDF5: %s is neither a group nor a dataset" % name)

    def close(self):
        if self._id is not None and self._owner:
            lib().H5Gclose(self._id)
        self._id = None

    def __del__(self):
        self.close()


class File(Group):
    def __init__(self, path, mode="r"):
        L = lib()
        p = os.fsencode(path)
        if mode == "r":
            fid = L.H5Fopen(p, H5F_ACC_RDONLY, 0)
        elif mode in ("r+", "a") and os.path.exists(path):
            fid = L.H5Fopen(p, H5F_ACC_RDWR, 0)
        elif mode in ("w", "a"):
            fid = L.H5Fcreate(p, H5F_ACC_TRUNC, 0, 0)
        else:
            raise ValueError("mode must be 'r', 'r+', 'w' or 'a'")
        if fid < 0:
            raise H5Error("HDF5: cannot open %s (mode %s)" % (path, mode))
        self._children = weakref.WeakSet()
        Group.__init__(self, fid, owner=False)
        self._file = self
        self._fid = fid

    def close(self):
        if getattr(self, "_fid", None) is not None:
            for child in list(self._children):
                child.close()
            lib().H5Fclose(self._fid)
            self._fid = None
            self._id = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        self.close()
