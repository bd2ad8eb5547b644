"""CPU pins of the Q-net's STRUCTURE and of the training call to the reference's own `alpha_nnet.py`:
tests/golden/net_graph.npz holds what `AlphaNNet.__init__`, `copy_and_compile`, `train` and `save` (alpha_nnet.py:10-56, 58-59,
78-109) build and call when run, unmodified, against recording stand-ins for the Keras names they import
(tests/golden/make_golden.py::record_net_graph).  Checked here: the weight list this build uses everywhere (order, shapes),
the functional graph it writes into its `.h5` files, which tensors carry the l2(1e-5) regularizer, the learning-rate
schedule, loss, fit arguments, save path, obstacle threshold.  The arithmetic BEHIND the recorded names (Conv2D, BatchNormalization
with Keras' default momentum 0.99 / epsilon 1e-3, Adam's defaults) is Keras' and stays unpinned: TensorFlow is absent."""
import inspect
import json

import numpy as np

from conftest import load_golden


def _graph():
    return json.loads(load_golden("net_graph.npz")["json"].tobytes().decode())


def _walk(graph):
    """tensor id -> shape, and the Keras get_weights() list (name, shape, regularized) the recorded layers imply"""
    shape, weights, producer = {}, [], {}
    for g in graph:
        kind, out = g["layer"], g["out"]
        if kind == "Input":
            shape[out] = tuple(g["args"][0])
        elif kind == "Conv2D":
            h, w, cin = shape[g["in"]["tensor"]]
            filters, (kh, kw) = g["args"]
            assert g["kwargs"]["use_bias"] is False
            assert g["kwargs"].get("padding", "valid") == "same" or (kh, kw) == (1, 1)      # 'valid' only where it changes nothing
            weights.append(("kernel", (kh, kw, cin, filters), g["kwargs"]["kernel_regularizer"]))
            shape[out] = (h, w, filters)
        elif kind == "BatchNormalization":
            assert g["kwargs"] == {"axis": 3} and g["args"] == []                         # channels last, every other setting Keras' default
            c = shape[g["in"]["tensor"]][2]
            weights += [(n, (c,), None) for n in ("gamma", "beta", "moving_mean", "moving_variance")]
            shape[out] = shape[g["in"]["tensor"]]
        elif kind == "Activation":
            shape[out] = shape[g["in"]["tensor"]]
        elif kind == "Add":
            a, b = (t["tensor"] for t in g["in"])
            assert shape[a] == shape[b]
            shape[out] = shape[a]
        elif kind == "Flatten":
            shape[out] = (int(np.prod(shape[g["in"]["tensor"]])),)
        elif kind == "Dense":
            (n_in,), units = shape[g["in"]["tensor"]], g["args"][0]
            weights += [("kernel", (n_in, units), g["kwargs"]["kernel_regularizer"]), ("bias", (units,), None)]
            shape[out] = (units,)
        if out is not None:
            producer[out] = g
    return shape, weights, producer


def test_weight_list_is_the_one_the_reference_constructor_implies():
    from snake_engine.net import glorot_uniform_weights
    d = _graph()
    shape, weights, _ = _walk(d["graph"])
    ws = glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    assert [tuple(w.shape) for w in ws] == [s for _, s, _ in weights]
    assert len(ws) == 54 and sum(int(np.prod(s)) for _, s, _ in weights) == 1_244_807
    model = [g for g in d["graph"] if g["layer"] == "Model"][0]
    assert shape[model["kwargs"]["outputs"]["tensor"]] == (3,) and shape[model["kwargs"]["inputs"]["tensor"]] == (21, 21, 3)


def test_l2_regularizer_sits_on_every_kernel_and_nowhere_else():
    import torch
    from snake_engine import train_step
    from snake_engine.net import glorot_uniform_weights
    from utils import trainer_torch
    _, weights, _ = _walk(_graph()["graph"])
    want = [i for i, (_, _, reg) in enumerate(weights) if reg is not None]
    for i in want:
        assert weights[i][0] == "kernel" and weights[i][2] == {"obj": "l2", "args": [1e-05], "kwargs": {}}
    assert all(reg is not None for name, _, reg in weights if name == "kernel")
    net = trainer_torch._Net(glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), torch.device("cpu"))
    assert sorted(net.kernel_idx) == want
    assert trainer_torch.L2_C == train_step.L2_C == 1e-05
    assert (trainer_torch.BN_MOMENTUM, trainer_torch.BN_EPS) == (train_step.BN_MOMENTUM, train_step.BN_EPS) == (0.99, 1e-3)   # Keras defaults


def test_graph_wiring_matches_the_functional_graph_written_into_h5_files():
    """utils.checkpoint.model_config (what `.h5` files carry as `model_config`) against the recorded constructor calls: the same
    layers in the same order with the same arguments, every Add fed by (second batch norm of the block, the block's input)"""
    from utils import checkpoint
    d = _graph()
    _, _, producer = _walk(d["graph"])
    cfg = checkpoint.model_config((21, 21, 3), 4)["config"]
    ours = [l for l in cfg["layers"]]
    rec = [g for g in d["graph"] if g["layer"] != "Model"]
    assert len(ours) == len(rec) == 40
    name_of = {}
    for l, g in zip(ours, rec):
        kind = {"Input": "InputLayer"}.get(g["layer"], g["layer"])
        assert l["class_name"] == kind, (l["name"], g)
        name_of[g["out"]] = l["name"]
        c = l["config"]
        if kind == "InputLayer":
            assert c["batch_input_shape"] == [None] + g["args"][0]
        elif kind == "Conv2D":
            assert (c["filters"], c["kernel_size"]) == (g["args"][0], g["args"][1]) and c["use_bias"] is False
            assert c["padding"] == g["kwargs"].get("padding", "valid") and c["strides"] == [1, 1] and c["activation"] == "linear"
            assert abs(c["kernel_regularizer"]["config"]["l2"] - g["kwargs"]["kernel_regularizer"]["args"][0]) < 1e-12      # float32(1e-5) in the file
        elif kind == "BatchNormalization":
            assert c["axis"] == [g["kwargs"]["axis"]] and (c["momentum"], c["epsilon"]) == (0.99, 0.001)
        elif kind == "Activation":
            assert c["activation"] == g["args"][0]
        elif kind == "Dense":
            assert c["units"] == g["args"][0] and c["use_bias"] is True and c["activation"] == "linear"
        ins = g["in"] if isinstance(g["in"], list) else ([g["in"]] if g["in"] else [])
        assert [n[0] for n in (l["inbound_nodes"][0] if l["inbound_nodes"] else [])] == [name_of[t["tensor"]] for t in ins]
    adds = [g for g in d["graph"] if g["layer"] == "Add"]
    assert len(adds) == 4
    for g in adds:
        bn_out, shortcut = (t["tensor"] for t in g["in"])
        assert producer[bn_out]["layer"] == "BatchNormalization"
        conv2 = producer[producer[bn_out]["in"]["tensor"]]
        relu1 = producer[conv2["in"]["tensor"]]
        bn1 = producer[relu1["in"]["tensor"]]
        conv1 = producer[bn1["in"]["tensor"]]
        assert (conv2["layer"], relu1["layer"], bn1["layer"], conv1["layer"]) == ("Conv2D", "Activation", "BatchNormalization", "Conv2D")
        assert conv1["in"]["tensor"] == shortcut and producer[shortcut]["layer"] == "Activation"     # the block's input, after its ReLU
    assert cfg["output_layers"][0][0] == name_of[[g for g in d["graph"] if g["layer"] == "Model"][0]["kwargs"]["outputs"]["tensor"]]


def test_compile_fit_save_and_obstacle_threshold():
    from utils import alpha_nnet, trainer_torch
    d = _graph()
    for c in d["compiled"] + [{"learning_rate": 0.0001, "calls": d["default_compile_and_calls"]}]:
        calls = {k[0]: k[1:] for k in c["calls"]}
        comp = calls["compile"][0]
        assert comp["loss"] == "mean_squared_error" and comp["optimizer"]["obj"] == "Adam" and comp["optimizer"]["args"] == []
        sched = comp["optimizer"]["kwargs"]["learning_rate"]
        assert list(comp["optimizer"]["kwargs"]) == ["learning_rate"] and sched["obj"] == "PiecewiseConstantDecay"     # Adam's other settings: Keras defaults
        boundaries, values = alpha_nnet.lr_schedule(c["learning_rate"])
        assert sched["args"] == [boundaries, values]                                # the same products, bit for bit
        assert values[-1] == 0.0 and trainer_torch.lr_at(100, (boundaries, values)) == values[4] and trainer_torch.lr_at(101, (boundaries, values)) == 0.0
    assert alpha_nnet.lr_schedule() == alpha_nnet.lr_schedule(0.0001)
    fits = [k for k in d["default_compile_and_calls"] if k[0] == "fit"]
    assert fits[0][2] == {"epochs": 32, "batch_size": 4} and fits[1][2] == {"epochs": 32, "batch_size": 2048}
    assert fits[0][1] == [[5, 21, 21, 3], "float32", [5, 3]]                         # array(X), array(Y) as they are
    sig = inspect.signature(alpha_nnet.AlphaNNet.train)
    assert (sig.parameters["epochs"].default, sig.parameters["batch_size"].default) == (32, 2048)
    assert inspect.signature(alpha_nnet.AlphaNNet.copy_and_compile).parameters["learning_rate"].default == 0.0001
    saves = [k for k in d["default_compile_and_calls"] if k[0] == "save"]
    assert saves == [["save", "models/g7.h5"]]
    seen = []

    class FakeVNet:
        def save(self, path):
            seen.append(path)
    nn = alpha_nnet.AlphaNNet()                                                       # no arguments: no net (alpha_nnet.py:10-13)
    assert nn.v_net is None
    nn.v_net = FakeVNet()
    nn.save("g7")
    assert seen == ["models/g7.h5"]
    for v, blocked in d["is_obstacle"]:
        assert bool(nn.is_obstacle(np.float32(v))) is blocked
