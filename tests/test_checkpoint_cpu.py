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
