"""CPU tests of the Keras-layout .h5 checkpoint I/O (utils/checkpoint.py)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest


def _weights(input_shape=(21, 21, 3), blocks=4, seed=0):
    from snake_engine.net import glorot_uniform_weights
    ws = glorot_uniform_weights(input_shape, blocks, seed)
    rng = np.random.RandomState(seed)
    return [w if w.ndim > 1 else (w + rng.randn(*w.shape).astype(np.float32) * 0.1) for w in ws]


@pytest.mark.parametrize("shape,blocks", [((21, 21, 3), 4), ((37, 37, 3), 10)])
def test_round_trip(tmp_path, shape, blocks):
    from utils import checkpoint
    ws = _weights(shape, blocks)
    path = str(tmp_path / "m1.h5")
    checkpoint.save_h5(path, ws, shape)
    got, ishape = checkpoint.load_h5(path)
    assert ishape == shape and len(got) == len(ws) == 14 + 10 * blocks
    for a, b in zip(ws, got):
        assert a.shape == b.shape and a.tobytes() == b.tobytes()


def test_missing_file_raises_oserror(tmp_path):
    from utils import checkpoint
    with pytest.raises(OSError):                      # pit.py:58 polls for the next generation on OSError
        checkpoint.load_h5(str(tmp_path / "nope.h5"))
    (tmp_path / "junk.h5").write_bytes(b"not an hdf5 file")
    with pytest.raises(OSError):
        checkpoint.load_h5(str(tmp_path / "junk.h5"))


def test_keras_layout(tmp_path):
    """the file has the groups/attributes/datasets Keras 2.x writes (checked with the HDF5 tools when present)"""
    from utils import checkpoint
    ws = _weights()
    path = str(tmp_path / "layout.h5")
    checkpoint.save_h5(path, ws, (21, 21, 3))
    plan, blocks = checkpoint.layer_plan(len(ws))
    assert blocks == 4 and [p[0] for p in plan][:4] == ["conv2d", "batch_normalization", "conv2d_1", "batch_normalization_1"]
    assert plan[-1][0] == "dense_1" and plan[-3][0] == "batch_normalization_9" and plan[-4][0] == "conv2d_9"
    cfg = checkpoint.model_config((21, 21, 3), 4)
    names = [l["name"] for l in cfg["config"]["layers"]]
    assert names[0] == "input_1" and names.count("flatten") == 1 and "add_3" in names and "activation_11" in names
    assert sum(l["class_name"] == "Conv2D" for l in cfg["config"]["layers"]) == 10
    json.dumps(cfg)
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        pytest.skip("h5dump not available")
    out = subprocess.run([h5dump, "-n", "1", path], capture_output=True, text=True).stdout
    for needle in ("/model_weights/conv2d/conv2d/kernel:0", "/model_weights/batch_normalization_9/batch_normalization_9/moving_variance:0",
                   "/model_weights/dense_1/dense_1/bias:0", "/model_weights/activation", "keras_version", "model_config",
                   "layer_names", "weight_names"):
        assert needle in out, needle
    hdr = subprocess.run([h5dump, "-H", "-d", "/model_weights/conv2d_1/conv2d_1/kernel:0", path], capture_output=True, text=True).stdout
    assert "( 3, 3, 128, 128 )" in hdr and "H5T_IEEE_F32LE" in hdr


# ---- a file shaped the way Keras 2.2.4-tf + h5py write it, made with libhdf5 directly (NOT with save_h5) ----------------------------
def _write_keras_style(path, ws, shape, vlen):
    """`Model.save(path)` of a compiled, trained functional model as keras/saving/hdf5_format.py (TF 2.1) lays it out, written call by
    call through libhdf5.  What save_h5 does not produce and a file from the reference's own stack has:
      * root attributes keras_version / backend / model_config / training_config as VARIABLE-LENGTH UTF-8 strings (h5py given `str`,
        h5py >= 3) when `vlen`, else as fixed-length byte strings (h5py 2.10 given `.encode('utf8')`, what TF 2.1 passes);
      * `weight_names` of the weightless layers (input_1, activation_*, add_*, flatten) as h5py stores `np.asarray([])`: an EMPTY
        FLOAT64 array attribute, not a string attribute;
      * fixed-length 'S' arrays for layer_names / weight_names, padded to the longest entry;
      * an `optimizer_weights` group (Adam's iteration count as an int64 scalar, the moments as float32 arrays) next to
        `model_weights`, and the layer groups created in model order."""
    import ctypes as C
    from utils import checkpoint as cp
    L = cp._lib()
    hid = cp.hid_t
    for name, res, args in [("H5Tset_cset", C.c_int, [hid, C.c_int])]:
        fn = getattr(L, name); fn.restype, fn.argtypes = res, args
    T_F64 = hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    T_I64 = hid.in_dll(L, "H5T_NATIVE_INT64_g").value
    ok = cp._ok

    def str_attr(obj, name, text):
        if not vlen:
            return cp._write_str_attr(L, obj, name, text)
        t = ok(L.H5Tcopy(L.T_STR), "H5Tcopy")
        L.H5Tset_size(t, C.c_size_t(-1).value)                 # H5T_VARIABLE
        L.H5Tset_cset(t, 1)                                    # H5T_CSET_UTF8
        sp = ok(L.H5Screate(cp.H5S_SCALAR), "H5Screate")
        a = ok(L.H5Acreate2(obj, name.encode(), t, sp, 0, 0), name)
        p = (C.c_char_p * 1)(text.encode("utf8"))
        ok(L.H5Awrite(a, t, p), name)
        L.H5Aclose(a); L.H5Sclose(sp); L.H5Tclose(t)

    def empty_f64_attr(obj, name):
        dims = (C.c_uint64 * 1)(0)
        sp = ok(L.H5Screate_simple(1, dims, None), "space")
        a = ok(L.H5Acreate2(obj, name.encode(), T_F64, sp, 0, 0), name)
        L.H5Aclose(a); L.H5Sclose(sp)

    def dataset(grp, name, arr, t):
        dims = (C.c_uint64 * max(1, arr.ndim))(*arr.shape)
        sp = ok(L.H5Screate_simple(arr.ndim, dims, None) if arr.ndim else L.H5Screate(cp.H5S_SCALAR), "space")
        ds = ok(L.H5Dcreate2(grp, name.encode(), t, sp, 0, 0, 0), name)
        ok(L.H5Dwrite(ds, t, 0, 0, 0, arr.ctypes.data), name)
        L.H5Dclose(ds); L.H5Sclose(sp)

    plan, blocks = cp.layer_plan(len(ws))
    cfg = cp.model_config(shape, blocks)
    with_w = {n: (wn, idx) for n, wn, idx in plan}
    names = [l["name"] for l in cfg["config"]["layers"]]
    training = {"loss": "mean_squared_error", "metrics": [], "weighted_metrics": None, "sample_weight_mode": None, "loss_weights": None,
                "optimizer_config": {"class_name": "Adam", "config": {"name": "Adam", "learning_rate": {"class_name": "PiecewiseConstantDecay",
                                     "config": {"boundaries": [20, 40, 60, 80, 100], "values": [1e-4, 2.5e-5, 6.25e-6, 1.5625e-6, 3.90625e-7, 0.0], "name": None}},
                                     "decay": 0.0, "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-07, "amsgrad": False}}}
    f = ok(L.H5Fcreate(path.encode(), cp.H5F_ACC_TRUNC, 0, 0), path)
    str_attr(f, "keras_version", "2.2.4-tf")
    str_attr(f, "backend", "tensorflow")
    str_attr(f, "model_config", json.dumps(cfg))
    g = ok(L.H5Gcreate2(f, b"model_weights", 0, 0, 0), "model_weights")
    cp._write_strlist_attr(L, g, "layer_names", names)
    str_attr(g, "backend", "tensorflow")
    str_attr(g, "keras_version", "2.2.4-tf")
    opt_names, opt_arrays = ["Adam/iter:0"], [np.array(100, np.int64)]
    for name in names:
        lg = ok(L.H5Gcreate2(g, name.encode(), 0, 0, 0), name)
        wn, idx = with_w.get(name, ([], []))
        if not wn:
            empty_f64_attr(lg, "weight_names")
        else:
            cp._write_strlist_attr(L, lg, "weight_names", [f"{name}/{w}" for w in wn])
            ig = ok(L.H5Gcreate2(lg, name.encode(), 0, 0, 0), name)
            for w, i in zip(wn, idx):
                dataset(ig, w, np.ascontiguousarray(ws[i], np.float32), L.T_FLOAT)
                if "moving" not in w:
                    for slot in ("m", "v"):
                        opt_names.append(f"Adam/{name}/{w[:-2]}/{slot}:0")
                        opt_arrays.append(np.full(ws[i].shape, 1e-3 if slot == "m" else 1e-6, np.float32))
            L.H5Gclose(ig)
        L.H5Gclose(lg)
    L.H5Gclose(g)
    str_attr(f, "training_config", json.dumps(training))
    og = ok(L.H5Gcreate2(f, b"optimizer_weights", 0, 0, 0), "optimizer_weights")
    cp._write_strlist_attr(L, og, "weight_names", opt_names)
    made = set()
    for n, a in zip(opt_names, opt_arrays):                    # 'Adam/conv2d/kernel/m:0': nested groups, as h5py makes them from the path
        parts, cur = n.split("/"), og
        opened = []
        for depth in range(len(parts) - 1):
            key = "/".join(parts[:depth + 1])
            cur = ok((L.H5Gopen2(cur, parts[depth].encode(), 0) if key in made else L.H5Gcreate2(cur, parts[depth].encode(), 0, 0, 0)), key)
            made.add(key); opened.append(cur)
        dataset(cur, parts[-1], a, T_I64 if a.dtype == np.int64 else L.T_FLOAT)
        for h in reversed(opened):
            L.H5Gclose(h)
    L.H5Gclose(og)
    L.H5Fclose(f)


@pytest.mark.parametrize("vlen", [True, False], ids=["h5py3-vlen-utf8", "h5py2-fixed-bytes"])
def test_reads_a_file_shaped_like_keras_own(tmp_path, vlen):
    """alpha_nnet.py:12 `load_model(model_name)` must accept what alpha_nnet.py:108-109 `v_net.save` of the reference's own stack wrote:
    load_h5 on a Keras / h5py shaped file (variable-length string attributes -- checkpoint.py's H5Tis_variable_str branch --, empty
    float64 weight_names, optimizer_weights, training_config) returns the same weights in get_weights() order and the input shape"""
    from utils import checkpoint
    shape = (21, 21, 3)
    ws = _weights(shape, 4, seed=3)
    path = str(tmp_path / ("keras_vlen.h5" if vlen else "keras_fixed.h5"))
    _write_keras_style(path, ws, shape, vlen)
    got, ishape = checkpoint.load_h5(path)
    assert ishape == shape and len(got) == len(ws) == 54
    for a, b in zip(ws, got):
        assert a.shape == b.shape and a.tobytes() == b.tobytes()
    # the attributes really are what the docstring says (so the branch under test was the one that ran)
    L = checkpoint._lib()
    f = L.H5Fopen(path.encode(), checkpoint.H5F_ACC_RDONLY, 0)
    a = L.H5Aopen(f, b"model_config", 0)
    t = L.H5Aget_type(a)
    assert (L.H5Tis_variable_str(t) > 0) == vlen
    L.H5Tclose(t); L.H5Aclose(a)
    g = L.H5Gopen2(f, b"model_weights", 0)
    lg = L.H5Gopen2(g, b"activation_3", 0)
    assert checkpoint._read_attr(L, lg, "weight_names") == []          # the empty float64 attribute of a weightless layer
    L.H5Gclose(lg); L.H5Gclose(g)
    assert L.H5Aexists(f, b"training_config") > 0
    og = L.H5Gopen2(f, b"optimizer_weights", 0)
    assert og >= 0 and len(checkpoint._read_attr(L, og, "weight_names")) == 1 + 2 * (10 + 2 * 10 + 4)      # iter + (m, v) per trainable tensor
    L.H5Gclose(og); L.H5Fclose(f)
    # what this build writes and the Keras-shaped file agree on everything load_h5 reads
    mine = str(tmp_path / "mine.h5")
    checkpoint.save_h5(mine, ws, shape)
    again, _ = checkpoint.load_h5(mine)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(again, got))


def test_truncated_keras_file_raises_oserror(tmp_path):
    """pit.py:58 polls `AlphaNNet(model_name)` while train.py may still be writing the next generation: a file cut short must raise
    OSError (what h5py / Keras raise), never return partial weights"""
    from utils import checkpoint
    ws = _weights((21, 21, 3), 4, seed=5)
    path = str(tmp_path / "cut.h5")
    _write_keras_style(path, ws, (21, 21, 3), True)
    blob = open(path, "rb").read()
    for keep in (len(blob) // 2, len(blob) - 4096, 600):
        open(path, "wb").write(blob[:keep])
        with pytest.raises(OSError):
            checkpoint.load_h5(path)
