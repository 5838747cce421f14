"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares
(no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from graphnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _header_functions():
    text = open(os.path.join(ROOT, "include", "graphnet_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gn_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from graphnet_amd import _lib
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/graphnet_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in graphnet_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_host_only_entry_points(lib):
    from graphnet_amd import _lib
    header = open(os.path.join(ROOT, "include", "graphnet_amd.h")).read()
    declared = int(re.search(r"#define\s+GN_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.gn_abi_version() == declared == _lib.ABI_VERSION == 4
    assert lib.gn_edge_slots(8) == 8 and lib.gn_edge_slots(9) == 16 and lib.gn_edge_slots(17) == 32
    assert lib.gn_scan_tmp_ints(150000) >= 74
    import ctypes
    w = (ctypes.c_int32 * 2)(32, 256)
    assert lib.gn_linear_wgrad_parts(1, 150000, 336, 2, ctypes.cast(w, ctypes.c_void_p)) >= 1 and lib.gn_colsum_blocks(1000) == 4
    assert lib.gn_edgeconv_dw2_slabs(0, 150_000, 8, 352, 256) >= 300
    assert lib.gn_edgeconv_saved_bytes(1000, 8, 352, 256) >= 1000 * 256 + 8000 * 44


def test_argument_validation_returns_error_codes_without_launching(lib):
    # bad shapes are rejected on the host before any kernel launch
    rc = lib.gn_knn_graph(None, 3, None, 3, None, None, 1, 10, 0, 0, None, None, None)
    assert rc != 0 and b"gn_knn_graph" in lib.gn_last_error()
    rc = lib.gn_edgeconv_fwd(1, None, None, None, None, 10, 8, None, 100, 100, None, None, 256, None, 256, None, None, 0,
                             None, None)
    assert rc != 0 and b"H1p%32" in lib.gn_last_error()
    rc = lib.gn_graph_globals(None, 7, 3, None, 1, None, None, 8, None, None, None)
    assert rc != 0


def test_code_object_targets_gfx950_only():
    from graphnet_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in data


def test_no_kernel_uses_scratch_memory():
    """Every kernel of the library must have a private-segment size of 0: a register spill costs a scratch round trip
    with a full ``vmcnt`` drain in the hot loops, and hipGraph replays of a training step that contained a kernel
    with scratch (``edge_fwd_ws_kernel<22,21,8>``, 48 bytes) ended in "Memory access fault ... write access to a
    read-only page" after a handful of replays; with the spill gone the same replays run clean."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("code_object_meta", os.path.join(root, "tools", "code_object_meta.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from graphnet_amd import _lib
    ks = mod.kernels(_lib.LIB_PATH)
    assert len(ks) > 100 and all("gfx950" in k["arch"] for k in ks)
    bad = [(k["name"], k["scratch"]) for k in ks if k["scratch"]]
    assert not bad, bad
