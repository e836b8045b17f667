"""CPU-side checks of the product library: it builds, loads, exports every symbol the
header declares, and refuses to compute without a GPU (no silent fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    from generalsreinforcementlearning_amd.csrc import build as B  # noqa
    return B.build(verbose=False)


def test_header_symbols_exported(built):
    import generalsreinforcementlearning_amd as g
    from generalsreinforcementlearning_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "generals_vec.h")).read()
    declared = set(re.findall(r"\b(gvec_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))
    L = g.load()
    for name in declared:
        assert hasattr(L, name), name
    assert L.gvec_abi_version() == 2


def test_config_defaults_match_reference(built):
    import ctypes as C
    import generalsreinforcementlearning_amd as g
    from generalsreinforcementlearning_amd._lib import Config
    cfg = Config()
    assert g.load().gvec_config_default(C.byref(cfg)) == 0
    # internal/config/config.go:206-209, internal/game/engine_initializer.go:118
    assert (cfg.prod_general, cfg.prod_city, cfg.prod_normal, cfg.normal_growth_interval, cfg.fog_of_war) == (1, 1, 1, 25, 1)


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import generalsreinforcementlearning_amd as g
    with pytest.raises(g.GvecError) as ei:
        g.VecEngine(4, 10, 10, 2)
    assert ei.value.code == -2  # GVEC_E_NO_DEVICE


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "generalsreinforcementlearning_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "generals_oracle" not in txt and "_oracle" not in txt, f


def test_cpp_host_mirror_compiles():
    import subprocess, tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".cpp", delete=False) as f:
        f.write('#include "generalsreinforcementlearning_amd/host/vec_engine.hpp"\nint main() { gvec::GameConfig c; (void)c; return 0; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-I", ROOT, f.name])
    os.unlink(f.name)


def test_sharded_create_fails_loudly_without_a_gpu():
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    import generalsreinforcementlearning_amd as g
    from generalsreinforcementlearning_amd._lib import Config
    L = g.load()
    cfg = Config()
    L.gvec_config_default(C.byref(cfg))
    cfg.num_envs = 8
    h = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 1)
    assert L.gvec_create_sharded(C.byref(cfg), devs, 2, C.byref(h)) == -2 and not h.value   # GVEC_E_NO_DEVICE
    assert b"no CPU fallback" in L.gvec_last_error()


def test_new_entry_points_validate_their_arguments_without_a_gpu():
    """Argument checks come before anything touches a device: they hold on a CPU-only box too."""
    import ctypes as C
    import generalsreinforcementlearning_amd as g
    L = g.load()
    bad = (C.c_int32 * 8)(10, 2, 4, 2, 2, 100, 0, 0)                      # record_dw far too small for this layout
    p = C.c_void_p(16)
    assert L.gvec_expand_experience_records(0, None, bad, p, 4, p, p, p, p) == -1 and b"layout" in L.gvec_last_error()
    assert L.gvec_expand_experience_records(0, None, bad, None, 4, p, p, p, p) == -1
    ok = (C.c_int32 * 8)(4 + 4 + (16 + 3) * 4 + 2 * 64, 2, 4, 2, 2, 100, 0, 0)
    assert L.gvec_expand_experience_records(0, None, ok, p, 0, p, p, p, p) == 0   # nothing to expand: no device needed
    assert L.gvec_stream_delta_cap(None) == -1 and L.gvec_num_shards(None) == -1
    from generalsreinforcementlearning_amd._lib import Config
    cfg = Config()
    L.gvec_config_default(C.byref(cfg))
    h = C.c_void_p()
    assert L.gvec_create_sharded(C.byref(cfg), None, 2, C.byref(h)) == -1
    assert L.gvec_create_sharded(C.byref(cfg), (C.c_int32 * 1)(0), 0, C.byref(h)) == -1
    out = C.c_void_p()
    assert L.gvec_host_alloc(0, C.byref(out)) == -1 and L.gvec_host_free(None) == 0


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's table says which reference interface each exported function replaces: none may be missing from it."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "generals_vec.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = [s for s in sorted(set(re.findall(r"\b(gvec_[a-z_0-9]+)\s*\(", hdr))) if s not in doc]
    assert not missing, missing
