"""The C-ABI library loads on a CPU-only box and exports every symbol include/yolact_hip.h
declares (no compute calls without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "yolact_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yh_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(built):
    from yolact_amd import capi
    L = capi.load_library()
    declared = _declared()
    assert len(declared) >= 30
    bound = {s[0] for s in capi.SYMBOLS}
    for name in declared:
        assert hasattr(L, name), f"{name} declared in yolact_hip.h but not exported"
        assert name in bound, f"{name} has no ctypes prototype"
    assert bound <= set(declared)


def test_version_and_defaults(built):
    from yolact_amd import capi
    L = capi.load_library()
    assert b"gfx950" in L.yh_version()
    cfg = capi.Config()
    L.yh_default_config(cfg)
    assert (cfg.abi_version, cfg.backbone, cfg.input_size, cfg.num_classes, cfg.top_k, cfg.max_dets) == (1, 50, 550, 81, 200, 100)
    assert abs(cfg.conf_thresh - 0.05) < 1e-7 and cfg.nms_thresh == 0.5


def test_no_cpu_fallback(built):
    """Without a GPU, creating an engine must fail loudly (never route through the oracle)."""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from yolact_amd import Engine, YhError
    with pytest.raises(YhError):
        Engine(input_size=128)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "tiny-object-detection_amd")
    for dp, _, fs in os.walk(pkg):
        if "build" in dp or "lib" in dp:
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "oracle.h" not in txt, os.path.join(dp, f)
                assert not re.search(r"\borc_[a-z0-9_]+\s*\(", txt), os.path.join(dp, f)  # no oracle call
    # tools/ are measurement scripts around the product: they do not load the oracle either
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "tools", f), errors="ignore").read()
            assert not re.search(r"import\s+(oracle|tfl_oracle)|liboracle|tfl_oracle as", txt), f
