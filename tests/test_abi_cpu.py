"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/wesup_hip.h declares, with the argument lists the ctypes binding uses.
No compute entry is called (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    hdr = open(os.path.join(ROOT, 'include', 'wesup_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return re.findall(r'\n\s*(?:int|long|size_t|const char\*|void\*)\s+(wesup_\w+)\s*\(([^;]*?)\)\s*;', hdr, flags=re.S)


@pytest.fixture(scope='module')
def lib():
    from wesup_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib


def test_library_exports_every_declared_symbol(lib):
    decls = _header_decls()
    assert len(decls) >= 30
    handle = lib.load()
    for name, _ in decls:
        assert hasattr(handle, name), f'{name} declared in include/wesup_hip.h but not exported'
    assert sorted(n for n, _ in decls) == lib.EXPORTS


def test_ctypes_signatures_match_header(lib):
    for name, args in _header_decls():
        sig = ''
        for a in [a.strip() for a in args.replace('\n', ' ').split(',')]:
            if a in ('void', ''):
                continue
            sig += ('p' if '*' in a else 'z' if a.startswith('size_t') else 'f' if a.startswith('float')
                    else 'l' if a.startswith('long') else 'i')
        assert lib._SIGS[name][1] == sig, name


def test_host_side_queries(lib):
    h = lib.load()
    assert h.wesup_abi_version() == lib.ABI_VERSION == 6
    assert h.wesup_conv3x3_kpad(3) == 64 and h.wesup_conv3x3_kpad(64) == 576 and h.wesup_conv3x3_kpad(512) == 4608
    assert h.wesup_strerror(0) == b'ok' and b'workspace' in h.wesup_strerror(-3)
    assert h.wesup_conv3x3_wgrad_workspace_bytes(4, 480, 480, 64, 64) > 0
    assert h.wesup_gemm_tn_workspace_bytes(1024, 2112, 2400) >= 1024 * 2112 * 4
    assert h.wesup_sp_preprocess_workspace_bytes(4, 480 * 480, 2, 640) > 0
    # invalid arguments are rejected on the host before any launch
    assert h.wesup_conv3x3_fwd(None, None, None, None, None, 1, 8, 8, 64, 64, 0, None, 0, None) == -1
    assert h.wesup_gemm_nt(None, 0, None, 0, None, None, 0, None, 0, 1, 1, 32, 0, None, 0, None) == -1
    # stream-K workspace: a short last round of 128x128 tiles (900 tiles on 512 slots) asks for partial-tile slots
    assert h.wesup_conv3x3_workspace_bytes(4, 120, 120, 256, 256) == 512 * 2 * 128 * 128 * 4
    assert h.wesup_gemm_nt_workspace_bytes(65536, 256, 2304) == 0          # 1024 tiles: two whole rounds


def test_product_path_fails_loudly_without_gpu_tensors(lib):
    import torch
    from wesup_amd import ops
    with pytest.raises(lib.WesupHipError):
        ops.pack_input(torch.zeros(1, 3, 4, 4))          # CPU tensor: no fallback


def test_winograd_host_side_queries_and_argument_checks(lib):
    """Sizes of the Winograd-domain operands (P x tiles x C floats each; P = (m+2)^2 positions, tiles = B * ceil(H/m) *
    ceil(W/m) for F(m x m, 3x3), m = 2 or 4) and the rejection of bad arguments on the host, before any launch."""
    h = lib.load()
    T = 4 * 30 * 30                                        # conv4_2 at the bench shape: 4 images of 60x60, 512 channels
    assert h.wesup_winograd_weight_floats(512, 512, 2) == 16 * 512 * 512
    assert h.wesup_winograd_weight_floats(512, 512, 4) == 36 * 512 * 512
    assert h.wesup_winograd_weight_floats(512, 512, 3) == 0                                                      # only F(2x2) and F(4x4)
    assert h.wesup_winograd_tiles(4, 60, 60, 2) == T and h.wesup_winograd_tiles(4, 60, 60, 4) == 4 * 15 * 15
    assert h.wesup_winograd_tiles(1, 7, 9, 4) == 2 * 3 and h.wesup_winograd_tiles(1, 7, 9, 8) == 0
    assert h.wesup_conv3x3_winograd_workspace_bytes(4, 60, 60, 512, 512, 2) == 2 * 16 * T * 512 * 4
    assert h.wesup_conv3x3_winograd_workspace_bytes(4, 60, 60, 512, 512, 4) == 2 * 36 * (T // 4) * 512 * 4      # 2.25x instead of 4x
    assert h.wesup_conv3x3_winograd_workspace_bytes(1, 7, 9, 256, 512, 2) == 16 * 4 * 5 * (256 + 512) * 4      # odd borders round up
    assert h.wesup_conv3x3_winograd_workspace_bytes(1, 7, 9, 256, 512, 4) == 36 * 2 * 3 * (256 + 512) * 4
    assert h.wesup_conv3x3_wgrad_winograd_workspace_bytes(4, 60, 60, 512, 512, 2) > 2 * 16 * T * 512 * 4       # + split-K slabs
    assert h.wesup_conv3x3_wgrad_winograd_workspace_bytes(4, 60, 60, 512, 512, 4) > 2 * 36 * (T // 4) * 512 * 4
    assert h.wesup_conv3x3_winograd_workspace_bytes(4, 60, 60, 3, 64, 2) == 0                                   # image layer: not for this path
    assert h.wesup_conv3x3_winograd_workspace_bytes(4, 60, 60, 512, 512, 3) == 0
    for m in (2, 4):
        assert h.wesup_conv3x3_fwd_winograd(None, None, None, None, None, None, 0, None, 4, 60, 60, 512, 512, 0, m, None, 0, None) == -1
        assert h.wesup_conv3x3_dgrad_winograd(None, None, None, None, 4, 60, 60, 512, 512, 0, m, None, 0, None) == -1
        assert h.wesup_conv3x3_wgrad_winograd(None, None, None, None, None, 4, 60, 60, 512, 512, 0, m, None, 0, None) == -1
        assert h.wesup_winograd_input_transform(None, None, 0, 1, 8, 8, 64, 0, m, None) == -1
        assert h.wesup_winograd_output_transform(None, 0, None, None, None, None, None, 0, 1, 8, 8, 64, 0, m, None) == -1
        assert h.wesup_winograd_outgrad_transform(None, None, None, 1, 8, 8, 64, m, None, 0, None) == -1
        assert h.wesup_winograd_filter_grad(None, 0, 0, 1, None, None, 64, 64, m, None) == -1
        assert h.wesup_winograd_pack_weight(None, None, None, 64, 64, m, None) == -1
    assert h.wesup_gemm_nt_batched(None, 0, 0, None, 0, 0, None, 0, 0, 16, 128, 128, 32, None) == -1
    assert h.wesup_scale_rows_by_area(None, None, 4, 64, None) == -1
    # the bias gradient of the F(4x4) weight gradient is summed inside the outgrad transform: per-block rows + colsum workspace
    assert h.wesup_winograd_outgrad_workspace_bytes(4, 60, 60, 512, 2) == 0
    blocks = -(-4 * 15 * 15 * 128 // 256)
    assert h.wesup_winograd_outgrad_workspace_bytes(4, 60, 60, 512, 4) == blocks * 512 * 4 + h.wesup_colsum_workspace_bytes(blocks, 512)
    # the debug entries only exist in `make debug`'s library: the shipped one refuses them
    import ctypes
    assert h.wesup_debug_clock(ctypes.byref(ctypes.c_double())) == -1 and h.wesup_debug_set_trace(None) == -1
    assert h.wesup_conv3x3_fwd_side(None, None, None, None, None, None, None, None, 32, 1, 8, 8, 64, 64, 0, None) == -1


def test_step_plan_object_without_a_gpu(lib):
    """The plan object itself is host code: create, record nothing, seal, compare, replay an empty range, destroy."""
    import ctypes
    h = lib.load()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    assert h.wesup_plan_create(ctypes.byref(a)) == 0 and h.wesup_plan_create(ctypes.byref(b)) == 0
    assert h.wesup_plan_replay(a, 0, 0) == -1                  # not sealed yet
    assert h.wesup_plan_begin(a) == 0
    assert h.wesup_plan_begin(b) == -1                         # one recording per thread at a time
    assert h.wesup_plan_end(b) == -1
    assert h.wesup_plan_size(a) == 0 and h.wesup_plan_end(a) == 0
    assert h.wesup_plan_begin(b) == 0 and h.wesup_plan_end(b) == 0
    assert h.wesup_plan_diff(a, b) == 0
    assert h.wesup_plan_replay(a, 0, 0) == 0 and h.wesup_plan_replay(a, 0, 1) == -1
    assert h.wesup_plan_kernels(a) == 0 and h.wesup_plan_node_name(a, 0) == b''
    assert h.wesup_sync_slots() >= 64
    assert h.wesup_plan_destroy(a) == 0 and h.wesup_plan_destroy(b) == 0
    assert h.wesup_winograd_pack_weights(None, 0, None) == -1 and h.wesup_transpose_batched(None, 3, None) == -1


def test_driver_build_entry_runs():
    """__graft_entry__.build() is what the driver calls on the CPU box every round: it must compile, load the library and agree
    with it about the ABI version (it asserted a stale version for most of round 4 while every other test was green)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    g = importlib.import_module('__graft_entry__')
    g.build()
