"""Keras-2.x-layout HDF5 checkpoints without h5py/TensorFlow (reference: AlphaNNet.save / load_model,
alpha_nnet.py:12, 108-109; file layout: SURVEY.md Appendix D.3).

The reference stores `models/<name><iter>.h5` with Keras `Model.save`: root attributes `keras_version`,
`backend`, `model_config` (JSON of the functional graph), group `model_weights` with attribute
`layer_names`, one group per layer with attribute `weight_names` and the datasets
`<layer>/<layer>/kernel:0` (conv: (kh,kw,in,out); dense: (in,out)), `bias:0`, and for BatchNormalization
`gamma:0, beta:0, moving_mean:0, moving_variance:0`.  This module writes and reads exactly that layout
through the HDF5 C library (libhdf5, ctypes).  No sample file exists in the reference, so the bytes are
unpinned; the layout follows Keras 2.2.4-tf (TF 2.1) `save_model_to_hdf5`.
"""
import ctypes as C
import ctypes.util
import json
import os

import numpy as np

_H5 = None
hid_t = C.c_int64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_SCALAR = 0
H5T_STR_NULLPAD = 1


def _lib():
    global _H5
    if _H5 is None:
        cands = [os.environ.get("SNK_LIBHDF5"), "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so",
                 ctypes.util.find_library("hdf5"), "libhdf5_serial.so.103", "libhdf5.so"]
        err = None
        for c in cands:
            if not c:
                continue
            try:
                L = C.CDLL(c)
                break
            except OSError as e:
                err = e
        else:
            raise OSError(f"libhdf5 not found (needed for .h5 checkpoints): {err}")
        L.H5open()
        for name, res, args in [
            ("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fopen", hid_t, [C.c_char_p, C.c_uint, hid_t]),
            ("H5Fclose", C.c_int, [hid_t]),
            ("H5Gcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), ("H5Gopen2", hid_t, [hid_t, C.c_char_p, hid_t]),
            ("H5Gclose", C.c_int, [hid_t]),
            ("H5Screate", hid_t, [C.c_int]), ("H5Screate_simple", hid_t, [C.c_int, C.c_void_p, C.c_void_p]),
            ("H5Sclose", C.c_int, [hid_t]), ("H5Sget_simple_extent_ndims", C.c_int, [hid_t]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
            ("H5Sget_simple_extent_npoints", C.c_int64, [hid_t]),
            ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            ("H5Dopen2", hid_t, [hid_t, C.c_char_p, hid_t]), ("H5Dclose", C.c_int, [hid_t]),
            ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Dread", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dget_space", hid_t, [hid_t]),
            ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), ("H5Aopen", hid_t, [hid_t, C.c_char_p, hid_t]),
            ("H5Aexists", C.c_int, [hid_t, C.c_char_p]),
            ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]), ("H5Aread", C.c_int, [hid_t, hid_t, C.c_void_p]),
            ("H5Aget_type", hid_t, [hid_t]), ("H5Aget_space", hid_t, [hid_t]), ("H5Aclose", C.c_int, [hid_t]),
            ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]), ("H5Tget_size", C.c_size_t, [hid_t]),
            ("H5Tset_strpad", C.c_int, [hid_t, C.c_int]), ("H5Tis_variable_str", C.c_int, [hid_t]), ("H5Tclose", C.c_int, [hid_t]),
            ("H5Dvlen_reclaim", C.c_int, [hid_t, hid_t, hid_t, C.c_void_p]),
            ("H5Eset_auto2", C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        ]:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        L.H5Eset_auto2(0, None, None)           # errors are reported through return codes -> Python exceptions
        L.T_FLOAT = hid_t.in_dll(L, "H5T_NATIVE_FLOAT_g").value
        L.T_STR = hid_t.in_dll(L, "H5T_C_S1_g").value
        _H5 = L
    return _H5


def _ok(v, what):
    if v < 0:
        raise OSError(f"HDF5: {what} failed")
    return v


def _str_type(L, n):
    t = _ok(L.H5Tcopy(L.T_STR), "H5Tcopy")
    L.H5Tset_size(t, max(1, n))
    L.H5Tset_strpad(t, H5T_STR_NULLPAD)
    return t


def _write_str_attr(L, obj, name, value):
    b = value if isinstance(value, bytes) else value.encode("utf8")
    t = _str_type(L, len(b))
    sp = _ok(L.H5Screate(H5S_SCALAR), "H5Screate")
    a = _ok(L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), f"create attribute {name}")
    buf = C.create_string_buffer(b, max(1, len(b)))
    _ok(L.H5Awrite(a, t, buf), f"write attribute {name}")
    L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)


def _write_strlist_attr(L, obj, name, values):
    vals = [v if isinstance(v, bytes) else v.encode("utf8") for v in values]
    n = max([len(v) for v in vals] + [1])
    t = _str_type(L, n)
    dims = (C.c_uint64 * 1)(len(vals))
    sp = _ok(L.H5Screate_simple(1, dims, None), "H5Screate_simple")
    a = _ok(L.H5Acreate2(obj, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), f"create attribute {name}")
    if vals:
        arr = np.array(vals, dtype=f"S{n}")
        _ok(L.H5Awrite(a, t, arr.ctypes.data), f"write attribute {name}")
    L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)


def _read_attr(L, obj, name):
    """returns bytes (scalar) or list of bytes (1-D), fixed or variable length strings"""
    a = _ok(L.H5Aopen(obj, name.encode(), H5P_DEFAULT), f"open attribute {name}")
    t = L.H5Aget_type(a)
    sp = L.H5Aget_space(a)
    npts = int(L.H5Sget_simple_extent_npoints(sp))
    nd = L.H5Sget_simple_extent_ndims(sp)
    try:
        if L.H5Tis_variable_str(t) > 0:
            ptrs = (C.c_char_p * max(1, npts))()
            _ok(L.H5Aread(a, t, ptrs), f"read attribute {name}")
            out = [ptrs[i] or b"" for i in range(npts)]
            L.H5Dvlen_reclaim(t, sp, H5P_DEFAULT, ptrs)
        else:
            sz = L.H5Tget_size(t)
            buf = C.create_string_buffer(max(1, sz * npts))
            if npts:
                _ok(L.H5Aread(a, t, buf), f"read attribute {name}")
            out = [buf.raw[i * sz:(i + 1) * sz].rstrip(b"\x00") for i in range(npts)]
    finally:
        L.H5Sclose(sp); L.H5Tclose(t); L.H5Aclose(a)
    return out[0] if nd == 0 else out


# ---------------------------------------------------------------------------------------------------
# the reference graph (alpha_nnet.py:19-56) as Keras layer names and a functional-model config
# ---------------------------------------------------------------------------------------------------
def layer_plan(n_weights):
    """[(layer_name, [weight names], [indices into the flat Keras weight list])] in Keras layer order"""
    blocks = (n_weights - 14) // 10
    plan, wi, ci, bi = [], 0, 0, 0

    def nm(base, k):
        return base if k == 0 else f"{base}_{k}"

    def conv_bn():
        nonlocal wi, ci, bi
        plan.append((nm("conv2d", ci), ["kernel:0"], [wi]))
        plan.append((nm("batch_normalization", bi), ["gamma:0", "beta:0", "moving_mean:0", "moving_variance:0"],
                     [wi + 1, wi + 2, wi + 3, wi + 4]))
        wi += 5; ci += 1; bi += 1
    for _ in range(1 + 2 * blocks + 1):
        conv_bn()
    plan.append(("dense", ["kernel:0", "bias:0"], [wi, wi + 1]))
    plan.append(("dense_1", ["kernel:0", "bias:0"], [wi + 2, wi + 3]))
    return plan, blocks


def model_config(input_shape, blocks):
    """Keras 2.2.4-tf functional `Model` config of the reference graph"""
    layers = []

    def add(cls, name, cfg, inbound):
        layers.append({"class_name": cls, "name": name, "config": dict(cfg, name=name),
                       "inbound_nodes": [[[i, 0, 0, {}] for i in inbound]] if inbound else []})
        return name
    reg = {"class_name": "L1L2", "config": {"l1": 0.0, "l2": 9.999999747378752e-06}}
    glorot = {"class_name": "GlorotUniform", "config": {"seed": None}}
    zeros, ones = {"class_name": "Zeros", "config": {}}, {"class_name": "Ones", "config": {}}
    cnt = {"conv2d": 0, "batch_normalization": 0, "activation": 0, "add": 0}

    def nm(base):
        k = cnt[base]; cnt[base] += 1
        return base if k == 0 else f"{base}_{k}"

    def conv(x, filters, ks, pad):
        return add("Conv2D", nm("conv2d"), {"trainable": True, "dtype": "float32", "filters": filters, "kernel_size": [ks, ks],
                                            "strides": [1, 1], "padding": pad, "data_format": "channels_last",
                                            "dilation_rate": [1, 1], "activation": "linear", "use_bias": False,
                                            "kernel_initializer": glorot, "bias_initializer": zeros, "kernel_regularizer": reg,
                                            "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None,
                                            "bias_constraint": None}, [x])

    def bn(x):
        return add("BatchNormalization", nm("batch_normalization"),
                   {"trainable": True, "dtype": "float32", "axis": [3], "momentum": 0.99, "epsilon": 0.001, "center": True,
                    "scale": True, "beta_initializer": zeros, "gamma_initializer": ones, "moving_mean_initializer": zeros,
                    "moving_variance_initializer": ones, "beta_regularizer": None, "gamma_regularizer": None,
                    "beta_constraint": None, "gamma_constraint": None}, [x])

    def act(x, fn):
        return add("Activation", nm("activation"), {"trainable": True, "dtype": "float32", "activation": fn}, [x])
    x = add("InputLayer", "input_1", {"batch_input_shape": [None] + list(input_shape), "dtype": "float32", "sparse": False,
                                      "ragged": False}, [])
    h = act(bn(conv(x, 128, 3, "same")), "relu")
    for _ in range(blocks):
        sc = h
        h = act(bn(conv(h, 128, 3, "same")), "relu")
        b2 = bn(conv(h, 128, 3, "same"))
        s = add("Add", nm("add"), {"trainable": True, "dtype": "float32"}, [b2, sc])
        h = act(s, "relu")
    h = act(bn(conv(h, 1, 1, "valid")), "relu")
    f = add("Flatten", "flatten", {"trainable": True, "dtype": "float32", "data_format": "channels_last"}, [h])

    def dense(x, name, units):
        return add("Dense", name, {"trainable": True, "dtype": "float32", "units": units, "activation": "linear", "use_bias": True,
                                   "kernel_initializer": glorot, "bias_initializer": zeros, "kernel_regularizer": reg,
                                   "bias_regularizer": None, "activity_regularizer": None, "kernel_constraint": None,
                                   "bias_constraint": None}, [x])
    h = act(dense(f, "dense", 128), "relu")
    y = act(dense(h, "dense_1", 3), "tanh")
    return {"class_name": "Model", "config": {"name": "model", "layers": layers, "input_layers": [["input_1", 0, 0]],
                                              "output_layers": [[y, 0, 0]]}}


def save_h5(path, weights, input_shape):
    L = _lib()
    plan, blocks = layer_plan(len(weights))
    cfg = model_config(input_shape, blocks)
    d = os.path.dirname(path)
    if d and not os.path.isdir(d):
        raise OSError(f"Unable to create file (unable to open file: name = '{path}', errno = 2, error message = "
                      "'No such file or directory')")
    f = _ok(L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), f"create {path}")
    try:
        _write_str_attr(L, f, "keras_version", "2.2.4-tf")
        _write_str_attr(L, f, "backend", "tensorflow")
        _write_str_attr(L, f, "model_config", json.dumps(cfg))
        g = _ok(L.H5Gcreate2(f, b"model_weights", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "create model_weights")
        with_w = {name: (wn, idx) for name, wn, idx in plan}
        names = [l["name"] for l in cfg["config"]["layers"]]
        _write_strlist_attr(L, g, "layer_names", names)
        _write_str_attr(L, g, "backend", "tensorflow")
        _write_str_attr(L, g, "keras_version", "2.2.4-tf")
        for name in names:
            lg = _ok(L.H5Gcreate2(g, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create group {name}")
            wn, idx = with_w.get(name, ([], []))
            _write_strlist_attr(L, lg, "weight_names", [f"{name}/{w}" for w in wn])
            if wn:
                ig = _ok(L.H5Gcreate2(lg, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create group {name}/{name}")
                for w, i in zip(wn, idx):
                    arr = np.ascontiguousarray(weights[i], np.float32)
                    dims = (C.c_uint64 * arr.ndim)(*arr.shape)
                    sp = _ok(L.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
                    ds = _ok(L.H5Dcreate2(ig, w.encode(), L.T_FLOAT, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create dataset {w}")
                    _ok(L.H5Dwrite(ds, L.T_FLOAT, 0, 0, H5P_DEFAULT, arr.ctypes.data), f"write dataset {w}")
                    L.H5Dclose(ds); L.H5Sclose(sp)
                L.H5Gclose(ig)
            L.H5Gclose(lg)
        L.H5Gclose(g)
    finally:
        L.H5Fclose(f)


def load_h5(path):
    """-> (weights in Keras get_weights() order, input_shape).  Raises OSError for a missing/invalid file,
    like keras.models.load_model (pit.py:58 polls on that)."""
    L = _lib()
    if not os.path.exists(path):
        raise OSError(f"SavedModel file does not exist at: {path}")
    f = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise OSError(f"Unable to open file (file signature not found): {path}")
    try:
        cfg = json.loads(_read_attr(L, f, "model_config").decode("utf8"))
        input_shape = None
        for l in cfg["config"]["layers"]:
            if l["class_name"] == "InputLayer":
                input_shape = tuple(l["config"]["batch_input_shape"][1:])
        g = _ok(L.H5Gopen2(f, b"model_weights", H5P_DEFAULT), "open model_weights")
        weights = []
        for name in _read_attr(L, g, "layer_names"):
            lg = _ok(L.H5Gopen2(g, name, H5P_DEFAULT), f"open group {name!r}")
            for wname in _read_attr(L, lg, "weight_names"):
                ds = _ok(L.H5Dopen2(lg, wname, H5P_DEFAULT), f"open dataset {wname!r}")
                sp = L.H5Dget_space(ds)
                nd = L.H5Sget_simple_extent_ndims(sp)
                dims = (C.c_uint64 * max(1, nd))()
                L.H5Sget_simple_extent_dims(sp, dims, None)
                arr = np.empty(tuple(int(dims[i]) for i in range(nd)), np.float32)
                _ok(L.H5Dread(ds, L.T_FLOAT, 0, 0, H5P_DEFAULT, arr.ctypes.data), f"read dataset {wname!r}")
                weights.append(arr)
                L.H5Sclose(sp); L.H5Dclose(ds)
            L.H5Gclose(lg)
        L.H5Gclose(g)
    finally:
        L.H5Fclose(f)
    if input_shape is None:
        raise OSError(f"{path}: no InputLayer in model_config")
    return weights, input_shape
