"""The C-ABI library loads on a CPU-only box and exports every symbol include/yolact_hip.h
declares (no compute calls without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(headers=("yolact_hip.h", "yolact_hip_debug.h")):
    """Every yh_* function the headers under include/ declare: the drop-in boundary (yolact_hip.h) and the measurement / study / test
    surface (yolact_hip_debug.h)."""
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(yh_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


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
    assert (cfg.abi_version, cfg.backbone, cfg.input_size, cfg.num_classes, cfg.top_k, cfg.max_dets) == (4, 50, 550, 81, 200, 100)
    assert abs(cfg.conf_thresh - 0.05) < 1e-7 and cfg.nms_thresh == 0.5
    assert cfg.precision == capi.PRECISION_F16 and all(v == -1 for v in cfg.tune.as_dict().values())


def test_struct_layouts_match_the_header(built, tmp_path):
    """The ctypes mirrors of yh_config / yh_tuning / yh_detection / yh_tensor_info have the C header's sizes and
    field offsets (checked with a C program compiled against include/yolact_hip.h)."""
    import ctypes as C
    import subprocess
    from yolact_amd import capi
    src = tmp_path / "layout.c"
    fields = ["tune.plan_cus", "tune.tfl_graph", "precision", "debug_tensors", "conf_thresh"]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "yolact_hip_debug.h"\nint main(void) {\n'
                   'printf("%zu %zu %zu %zu", sizeof(yh_config), sizeof(yh_tuning), sizeof(yh_detection), sizeof(yh_tensor_info));\n'
                   + "".join(f'printf(" %zu", offsetof(yh_config, {f}));\n' for f in fields) + "return 0; }\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(capi.Config), C.sizeof(capi.Tuning), C.sizeof(capi.Detection), C.sizeof(capi.TensorInfo),
            capi.Config.tune.offset + capi.Tuning.plan_cus.offset, capi.Config.tune.offset + capi.Tuning.tfl_graph.offset,
            capi.Config.precision.offset, capi.Config.debug_tensors.offset, capi.Config.conf_thresh.offset]
    assert got == want


def test_library_reads_no_environment_variable(built):
    """'No global state' (include/yolact_hip.h): every measurement switch is a field of yh_tuning on the handle; the
    product sources never call getenv and the built library does not even import it."""
    import subprocess
    from yolact_amd import capi
    csrc = os.path.join(ROOT, "tiny-object-detection_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f), errors="ignore").read(), f
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", capi.lib_path()], text=True)
    assert "getenv" not in undefined


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


def test_standin_librccl_builds_and_exports_what_the_library_binds(tmp_path):
    """tests/rccl_standin/librccl_standin.c (TEST infrastructure: the stand-in that lets a one-GPU box execute the n > 1 collective
    code, tests/test_gpu_rccl_standin.py) compiles with gcc against the ROCm headers and exports exactly the eight entry points
    csrc/engine.hip binds by name - and the product never names it: libyolact_hip.so still asks the loader for librccl.so.1."""
    import subprocess
    src = os.path.join(ROOT, "tests", "rccl_standin", "librccl_standin.c")
    so = tmp_path / "librccl.so.1"
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-I/opt/rocm/include", "-o", str(so), src, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    want = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclBroadcast", "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"}
    assert want <= have, want - have
    eng = open(os.path.join(ROOT, "tiny-object-detection_amd", "csrc", "engine.hip")).read()
    for sym in want:
        assert f'sym("{sym}")' in eng, sym
    assert "rccl_standin" not in eng.replace("tests/rccl_standin", "") and '"librccl.so.1"' in eng


def test_the_host_facing_header_stands_alone_and_carries_no_lab_notebook(tmp_path):
    """include/yolact_hip.h - what a maintainer binds - compiles on its own as C, keeps yh_tuning opaque (32 ints: the
    same size as the named struct of yolact_hip_debug.h), and declares no measurement, study, test or single-op entry point; the C++ host
    mirror includes only it."""
    import subprocess
    pub = open(os.path.join(ROOT, "include", "yolact_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", pub, flags=re.S)
    assert "int32_t knob[32]" in code and "plan_cus" not in code and "yolact_hip_debug.h" not in code
    for name in _declared(("yolact_hip.h",)):
        assert not name.startswith(("yh_debug_", "yh_op_", "yh_profile_")) and name not in ("yh_set_tuning", "yh_get_tuning", "yh_time_steps", "yh_scene_time", "yh_tfl_create_tuned"), name
    src = tmp_path / "pub.c"
    src.write_text('#include "yolact_hip.h"\ntypedef char ok[sizeof(yh_tuning) == 128 ? 1 : -1];\nint main(void) { yh_config c; yh_default_config(&c); return c.abi_version == YH_ABI_VERSION ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    for f in ("yolact.hpp", "yolact.cpp", "yolact_demo.cpp"):
        txt = open(os.path.join(ROOT, "tiny-object-detection_amd", "host", f)).read()
        assert "yolact_hip_debug.h" not in txt, f
