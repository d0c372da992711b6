"""ctypes binding of libwesup_hip.so (C ABI in include/wesup_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a
kernel entry fails, this module raises.  ``build()`` compiles it with hipcc for
gfx950 (cross-compiles without a GPU).
"""
import ctypes
import functools
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
# WESUP_HIP_LIB: another build of the same library (A/B measurements of kernel variants in one GPU session)
LIB_PATH = os.environ.get('WESUP_HIP_LIB') or os.path.join(CSRC, 'libwesup_hip.so')

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t

# name -> (restype, [argtypes]);  'p' pointer, 'i' int, 'f' float, 'z' size_t, 'l' long
_SIGS = {
    'wesup_abi_version': (c_int, ''),
    'wesup_debug_clock': (c_int, 'p'),
    'wesup_debug_set_trace': (c_int, 'p'),
    'wesup_strerror': (ctypes.c_char_p, 'i'),
    'wesup_augment': (c_int, 'pppppiiippiiiip'),
    'wesup_appearance_workspace_bytes': (c_size_t, 'iii'),
    'wesup_appearance': (c_int, 'pppiiipzp'),
    'wesup_pack_input': (c_int, 'ppiiip'),
    'wesup_conv3x3_kpad': (c_int, 'i'),
    'wesup_pack_conv3x3_weight': (c_int, 'pppiip'),
    'wesup_transpose': (c_int, 'ppiip'),
    'wesup_transpose_batched': (c_int, 'pip'),
    'wesup_scale_rows_by_area': (c_int, 'pplip'),
    'wesup_conv3x3_workspace_bytes': (c_size_t, 'iiiii'),
    'wesup_conv3x3_fwd': (c_int, 'pppppiiiiiipzp'),
    'wesup_conv3x3_fwd_side': (c_int, 'ppppppppiiiiiiip'),
    'wesup_conv3x3_dgrad': (c_int, 'ppppiiiiiipzp'),
    'wesup_conv3x3_wgrad_workspace_bytes': (c_size_t, 'iiiii'),
    'wesup_conv3x3_wgrad': (c_int, 'ppppiiiiiipzp'),
    'wesup_conv3x3_wgrad_winograd_workspace_bytes': (c_size_t, 'iiiiii'),
    'wesup_conv3x3_wgrad_winograd': (c_int, 'pppppiiiiiiipzp'),
    'wesup_winograd_weight_floats': (c_size_t, 'iii'),
    'wesup_winograd_tiles': (ctypes.c_long, 'iiii'),
    'wesup_winograd_pack_weight': (c_int, 'pppiiip'),
    'wesup_winograd_pack_weights': (c_int, 'pip'),
    'wesup_conv3x3_winograd_workspace_bytes': (c_size_t, 'iiiiii'),
    'wesup_conv3x3_fwd_winograd': (c_int, 'ppppppipiiiiiiipzp'),
    'wesup_conv3x3_dgrad_winograd': (c_int, 'ppppiiiiiiipzp'),
    'wesup_conv3x3_dgrad_winograd_unpool': (c_int, 'ppppiiiiiiiipzp'),
    'wesup_winograd_output_transform_unpool': (c_int, 'plppppiiiiiiip'),
    'wesup_winograd_fused_supported': (c_int, 'iii'),
    'wesup_winograd_fused_route': (c_int, 'iiil'),
    'wesup_winograd_set_fused_min_blocks': (c_int, 'i'),
    'wesup_winograd_gemm_output_transform': (c_int, 'plpppppippiiiiiiiip'),
    'wesup_winograd_gemm_output_transform_gather': (c_int, 'plppppiipppiiiiiip'),
    'wesup_winograd_input_transform_bits': (c_int, 'pplpiiiiip'),
    'wesup_winograd_bias_rows': (ctypes.c_long, 'iiii'),
    'wesup_winograd_dual_transform': (c_int, 'ppppiiiip'),
    'wesup_conv3x3_wgrad_winograd_pre': (c_int, 'pppipp' + 'iiiii' + 'pzp'),
    'wesup_winograd_gemm_output_transform_ex': (c_int, 'plpppppp' + 'i' + 'pppp' + 'ii' + 'ppp' + 'iiiiiii' + 'p'),
    'wesup_conv3x3_dgrad_winograd_gather': (c_int, 'ppppppppiiiiiiiipzp'),
    'wesup_winograd_input_transform': (c_int, 'ppliiiiiip'),
    'wesup_gemm_nt_batched': (c_int, 'pilpilpiliiiip'),
    'wesup_gemm_nt_batched_bias': (c_int, 'pilpilplpiliiiip'),
    'wesup_winograd_output_transform': (c_int, 'plpppppiiiiiiip'),
    'wesup_winograd_outgrad_workspace_bytes': (c_size_t, 'iiiii'),
    'wesup_winograd_outgrad_transform': (c_int, 'pppiiiiipzp'),
    'wesup_winograd_filter_grad': (c_int, 'pllippiiip'),
    'wesup_gemm_nt_workspace_bytes': (c_size_t, 'iii'),
    'wesup_gemm_nt': (c_int, 'pipippipiiiiipzp'),
    'wesup_gemm_tn_workspace_bytes': (c_size_t, 'iii'),
    'wesup_gemm_tn': (c_int, 'pipipipiiiipzp'),
    'wesup_gemm_tn_batched_workspace_bytes': (c_size_t, 'iiii'),
    'wesup_gemm_tn_batched': (c_int, 'pilpilpiliiiiipzp'),
    'wesup_colsum_workspace_bytes': (c_size_t, 'ii'),
    'wesup_colsum': (c_int, 'pipiipzp'),
    'wesup_maxpool2_fwd': (c_int, 'ppiiiiip'),
    'wesup_maxpool2_bwd': (c_int, 'pppiiiiip'),
    'wesup_upsample_fwd': (c_int, 'ppiiiiiiiip'),
    'wesup_upsample_bwd': (c_int, 'ppppiiiiiiiiip'),
    'wesup_upsample_bwd_group': (c_int, 'ppppppiiiippiiiiiip'),
    'wesup_sp_preprocess_workspace_bytes': (c_size_t, 'iiii'),
    'wesup_sp_preprocess': (c_int, 'ppiiii' + 'pppppppppp' + 'ppi' + 'pzp'),
    'wesup_spmaps_to_labels': (c_int, 'ppiip'),
    'wesup_sp_max_units': (c_int, 'ii'),
    'wesup_sp_segments': (c_int, 'piiippp'),
    'wesup_sp_pool_workspace_bytes': (c_size_t, 'iii'),
    'wesup_sp_pool_fwd': (c_int, 'ppppppiiiiiipzp'),
    'wesup_sp_pool_bwd': (c_int, 'ppppiiiiip'),
    'wesup_sp_pool_upsample_fwd': (c_int, 'ppppppiiiiiiiiiipzp'),
    'wesup_sp_interp_matrix': (c_int, 'pppiiiiiip'),
    'wesup_paint_fwd': (c_int, 'pppiiiiip'),
    'wesup_slic_num_centers': (c_int, 'iii'),
    'wesup_slic_workspace_bytes': (c_size_t, 'iiii'),
    'wesup_slic': (c_int, 'pppiiiifiifpzp'),
    'wesup_classifier_fwd': (c_int, 'ppppiip'),
    'wesup_classifier_bwd_workspace_bytes': (c_size_t, 'ii'),
    'wesup_classifier_bwd': (c_int, 'ppppppppiipzp'),
    'wesup_propagate': (c_int, 'ppppfipppiiiip'),
    'wesup_loss_fwd': (c_int, 'ppppffppiiip'),
    'wesup_loss_bwd': (c_int, 'ppppppffpiiip'),
    'wesup_head_fwd': (c_int, 'pppppppfipppiiiip'),
    'wesup_head_bwd': (c_int, 'pppppppffpppiiiipzp'),
    'wesup_classifier_bwd_finish': (c_int, 'pzppiip'),
    'wesup_cross_entropy_fwd': (c_int, 'pppfpiip'),
    'wesup_cross_entropy_bwd': (c_int, 'pppppfpiip'),
    'wesup_sgd_step': (c_int, 'pppzffffip'),
    'wesup_seg_metrics_workspace_bytes': (c_size_t, 'i'),
    'wesup_seg_metrics': (c_int, 'pppiiipzp'),
    # entries by the names of SURVEY.md 8(b) (csrc/named.hip)
    'wesup_sp_stats': (c_int, 'ppiiiipppp'),
    'wesup_conv1x1_workspace_bytes': (c_size_t, 'iii'),
    'wesup_conv1x1_fwd': (c_int, 'ppppiiipzp'),
    'wesup_conv1x1_dgrad': (c_int, 'pppiiiipzp'),
    'wesup_conv1x1_wgrad': (c_int, 'ppppiiipzp'),
    'wesup_linear_workspace_bytes': (c_size_t, 'iii'),
    'wesup_linear_fwd': (c_int, 'ppppiiiipzp'),
    'wesup_linear_bwd': (c_int, 'pppppppiiipzp'),
    'wesup_upsample_bilinear_ac_fwd': (c_int, 'ppiiiiiiiip'),
    'wesup_upsample_bilinear_ac_bwd': (c_int, 'ppiiiiiiiip'),
    'wesup_softmax_ce_fwd': (c_int, 'pppfppiip'),
    'wesup_softmax_ce_bwd': (c_int, 'pppppfpiip'),
    # step plans (csrc/plan.hip)
    'wesup_plan_create': (c_int, 'p'),
    'wesup_plan_destroy': (c_int, 'p'),
    'wesup_plan_begin': (c_int, 'p'),
    'wesup_plan_end': (c_int, 'p'),
    'wesup_plan_size': (c_int, 'p'),
    'wesup_plan_kernels': (c_int, 'p'),
    'wesup_plan_replay': (c_int, 'pii'),
    'wesup_plan_diff': (c_int, 'pp'),
    'wesup_plan_node_name': (ctypes.c_char_p, 'pi'),
    'wesup_plan_node_stream': (c_void_p, 'pi'),
    'wesup_plan_node_host_ns': (c_int, 'pip'),
    'wesup_sync_slots': (c_int, ''),
    'wesup_sync_record': (c_int, 'ip'),
    'wesup_sync_wait': (c_int, 'ip'),
    'wesup_sync_synchronize': (c_int, 'i'),
    'wesup_sync_query': (c_int, 'i'),
    'wesup_copy_to_host': (c_int, 'ppzp'),
    'wesup_copy': (c_int, 'ppzp'),
    'wesup_fill_words': (c_int, 'pizp'),
}
_T = {'p': c_void_p, 'i': c_int, 'f': c_float, 'z': c_size_t, 'l': ctypes.c_long}

EXPORTS = sorted(_SIGS)
ABI_VERSION = 6          # include/wesup_hip.h; a stale libwesup_hip.so with other signatures must not be called

_lib = None


class WesupHipError(RuntimeError):
    pass


class WinoFilter(ctypes.Structure):            # WesupWinoFilter (include/wesup_hip.h)
    _fields_ = [('w', c_void_p), ('u_fwd', c_void_p), ('u_dgrad', c_void_p), ('Cout', c_int), ('Cin', c_int)]


class TransposeItem(ctypes.Structure):         # WesupTransposeItem
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('rows', c_int), ('cols', c_int)]


def build(verbose=False):
    """Compile libwesup_hip.so for gfx950 in-tree (wesup_amd/csrc)."""
    cmd = ['make', '-C', CSRC, '-j4']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise WesupHipError('building libwesup_hip.so failed')
    return LIB_PATH


def load():
    """Load the library (no GPU needed to load; kernels need one to run)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WesupHipError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                            '(there is no CPU fallback for the HIP path)')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = [_T[a] for a in args]
    # pure size / shape queries (integers in, an integer out): answered from a cache after the first call -- a training step asks
    # ~150 of them, the same ones every step
    for name in _SIGS:
        if name.endswith('_bytes') or name in ('wesup_winograd_tiles', 'wesup_winograd_fused_supported', 'wesup_winograd_bias_rows',
                                               'wesup_conv3x3_kpad', 'wesup_sp_max_units'):
            if all(a in 'il' for a in _SIGS[name][1]):
                setattr(lib, name, functools.lru_cache(maxsize=4096)(getattr(lib, name)))
    if lib.wesup_abi_version() != ABI_VERSION:
        raise WesupHipError(f'{LIB_PATH} has ABI version {lib.wesup_abi_version()}, this package binds version '
                            f'{ABI_VERSION}: rebuild it (make -C wesup_amd/csrc)')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().wesup_strerror(rc).decode()
        raise WesupHipError(f'{what} failed: {msg} (code {rc})')


def call(name, *args):
    """Call an int-returning entry and raise on a non-zero code."""
    check(getattr(load(), name)(*args), name)
