"""The shipped gfx950 library loads on a CPU-only box and exports every symbol of include/mldhip.h."""
import ctypes
import os
import re

import pytest

from mld_hip import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(REPO, "include", "mldhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mldhip_[a-z0-9_]+)\s*\(", text)))


def test_binding_covers_header():
    assert _header_symbols() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.DEFAULT_LIB):
        pytest.skip("libmldhip.so not built yet (python __graft_entry__.py)")
    lib = ctypes.CDLL(_lib.DEFAULT_LIB)
    for name in _header_symbols():
        assert hasattr(lib, name), name
    assert lib.mldhip_abi_version() == _lib.ABI_VERSION


def test_library_contains_gfx950_code_object():
    if not os.path.exists(_lib.DEFAULT_LIB):
        pytest.skip("libmldhip.so not built yet")
    blob = open(_lib.DEFAULT_LIB, "rb").read()
    assert b"gfx950" in blob
    for kernel in (b"gemm_kernel", b"gemm_tile32_kernel", b"attn_decode_kernel", b"den_final_step_kernel", b"feats2joints_kernel"):
        assert kernel in blob


def test_no_device_fails_loudly():
    """No CPU fallback: creating an engine without an MI355X is an error, not a silent slow path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    if not os.path.exists(_lib.DEFAULT_LIB):
        pytest.skip("libmldhip.so not built yet")
    with pytest.raises(_lib.MldHipError) as ei:
        _lib.Engine()
    assert ei.value.code == -5      # MLDHIP_ENODEV


def test_product_never_imports_oracle_or_simulator():
    pkg = os.path.join(REPO, "motion-latent-diffusion_amd", "mld_hip")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f
            assert "libmldhip_sim" not in src and "hipemu" not in src, f


def test_every_option_of_set_option_is_documented_in_the_header_and_vice_versa():
    """mldhip_set_option's names (csrc/mldhip.hip) == the names the header documents, and each has a default in engine/state.hpp."""
    import re
    src = open(os.path.join(REPO, "motion-latent-diffusion_amd", "csrc", "mldhip.hip")).read()
    body = src.split("int mldhip_set_option(mldhip_handle* e, const char* name, int64_t value) {")[1].split("\n}\n")[0]
    code = set(re.findall(r'n == "([a-z0-9_]+)"', body))
    hdr = open(os.path.join(REPO, "include", "mldhip.h")).read()
    doc = set(re.findall(r'^ \*   "([a-z0-9_]+)"', hdr.split("int mldhip_set_option(")[0], re.M))
    assert code and code == doc, (sorted(code - doc), sorted(doc - code))
    state = open(os.path.join(REPO, "motion-latent-diffusion_amd", "csrc", "engine", "state.hpp")).read()
    named = set(re.findall(r'//\s*"([a-z0-9_]+)"', state))
    assert code <= named, sorted(code - named)
