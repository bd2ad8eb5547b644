"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares (no compute calls: there is no GPU here), and the product path refuses to run
without a GPU instead of falling back to CPU code."""
import os
import re

import pytest

from conftest import REPO


def header_symbols():
    names = set()
    for fn in os.listdir(os.path.join(REPO, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(REPO, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(snk_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    import snake_engine
    from snake_engine import _lib
    L = snake_engine.lib()
    syms = header_symbols()
    assert len(syms) >= 15
    for name in sorted(syms):
        assert hasattr(L, name), f"libsnake_engine.so lacks {name}"
        assert name in _lib.PROTOTYPES, f"no ctypes prototype for {name}"
    assert set(_lib.PROTOTYPES) <= syms, set(_lib.PROTOTYPES) - syms
    assert L.snk_version() == _lib.ABI_VERSION == int(re.search(r"#define SNK_ABI_VERSION (\d+)", open(os.path.join(REPO, "include", "snake_engine.h")).read()).group(1))


def test_no_cpu_fallback():
    import torch
    import snake_engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(snake_engine.EngineError):
        snake_engine.Engine(4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "alphasnake-zero_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(root, fn), errors="replace").read()
                assert "oracle" not in txt.replace("oracle/obs_key.py", "").lower() or fn.endswith(".md"), \
                    f"{os.path.join(root, fn)} mentions the oracle"
