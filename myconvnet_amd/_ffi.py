"""ctypes binding of libmcn_hip.so (declared in include/mcn.h).

The product path has no CPU fallback: if the shared library is missing, importing this module
raises and every block method fails loudly.  Build with `python myconvnet_amd/build.py`.
"""
import ctypes
import os

import torch  # noqa: F401  MUST precede loading libmcn_hip.so: both then share the one HIP runtime torch ships
#                          (pointers / streams from torch are only valid in that runtime instance)
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCN_LIB_PATH: load another build of the library (A/B runs of compile-time variants; no freshness check then).  This is also the ONLY
# way libmcn_cpu.so (the C-ABI as plain loops: csrc_cpu/mcn_cpu.cpp, test infrastructure for the host code) gets loaded: there is no
# fallback to it — without MCN_LIB_PATH a missing or stale libmcn_hip.so is an ImportError.
LIB_PATH = os.environ.get('MCN_LIB_PATH') or os.path.join(_HERE, 'libmcn_hip.so')

F32, BF16, F16 = 0, 1, 2
NHWC, NCHW = 0, 1
ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID, ACT_RELU6, ACT_LRELU, ACT_TANH = 0, 1, 2, 3, 4, 5, 6
DECAY_L2, DECAY_L1, DECAY_HUBER = 0, 1, 2
CONV_FWD, CONV_DGRAD, CONV_WGRAD = 0, 1, 2
OK, E_BADARG, E_UNSUPPORTED, E_LAUNCH, E_WORKSPACE = 0, -1, -2, -3, -4


class McnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('mcn error {}: {}'.format(code, msg))
        self.code = code


class ConvGeom(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ('N', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'SH', 'SW', 'DH', 'DW',
                                       'padT', 'padB', 'padL', 'padR', 'x_cs', 'tile')]


class PackJob(ctypes.Structure):
    _fields_ = [('w_hwio', c_void_p), ('packed', c_void_p), ('geom', ConvGeom), ('op', c_int32), ('reserved', c_int32)]


# name -> (restype, argtypes); every symbol include/mcn.h declares
SIGNATURES = {
    'mcn_version': (c_int, []),
    'mcn_last_error': (c_char_p, []),
    'mcn_build_id': (c_char_p, []),
    'mcn_conv2d_tile_candidates': (c_int, [c_int]),
    'mcn_conv2d_workspace_bytes': (c_size_t, [c_int, ctypes.POINTER(ConvGeom), c_int]),
    'mcn_conv2d_kslices': (c_int32, [c_int, ctypes.POINTER(ConvGeom), c_int]),
    'mcn_conv2d_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_conv2d_dgrad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_conv2d_dgrad_addmasked_ok': (c_int32, [ctypes.POINTER(ConvGeom), c_int]),
    'mcn_conv2d_dgrad_addmasked': (c_int, [c_void_p] * 6 + [ctypes.POINTER(ConvGeom), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_conv2d_packed_bytes': (c_size_t, [c_int, ctypes.POINTER(ConvGeom), c_int]),
    'mcn_conv2d_pack_table_bytes': (c_size_t, [ctypes.POINTER(PackJob), c_int32]),
    'mcn_conv2d_pack_table_build': (c_int, [ctypes.POINTER(PackJob), c_int32, c_int, c_void_p, c_size_t, ctypes.POINTER(c_int32)]),
    'mcn_conv2d_pack_run': (c_int, [c_void_p, c_int32, c_int, c_void_p]),
    'mcn_conv2d_kernel_name': (c_int, [c_int, ctypes.POINTER(ConvGeom), c_int, ctypes.c_char_p, c_size_t]),
    'mcn_conv2d_launch_list': (c_int, [c_int, ctypes.POINTER(ConvGeom), c_int, ctypes.c_char_p, c_size_t]),
    'mcn_conv2d_wgrad': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_workspace_bytes': (c_size_t, [c_int64, c_int32]),
    'mcn_bn_relu_mask_bytes': (c_size_t, [c_int64, c_int32, c_int]),
    'mcn_bn_fwd_train': (c_int, [c_void_p] * 12 + [c_float, c_int64, c_int32, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_fwd_train_gap': (c_int, [c_void_p] * 11 + [c_float, c_int32, c_int32, c_int32, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_conv2d_pair_geom': (c_int, [ctypes.POINTER(ConvGeom), c_int, ctypes.POINTER(ConvGeom)]),
    'mcn_conv2d_pair_weights': (c_int, [c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int, c_void_p]),
    'mcn_conv2d_pair_wgrad_fold': (c_int, [c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int, c_void_p]),
    'mcn_conv2d_bnstats_rows': (c_int32, [ctypes.POINTER(ConvGeom), c_int, ctypes.POINTER(c_int32)]),
    'mcn_conv2d_fwd_bnstats': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_fwd_train_fused_stats': (c_int, [c_void_p, c_int32, c_int32] + [c_void_p] * 8 + [c_float, c_int64, c_int32, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_fwd_train_fused_affskip': (c_int, [c_void_p, c_void_p, c_int32, c_int32] + [c_void_p] * 12 + [c_float, c_int64, c_int32, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_conv2d_dgrad_bnred_rows': (c_int32, [ctypes.POINTER(ConvGeom), c_int]),
    'mcn_conv2d_dgrad_bnred': (c_int, [c_void_p] * 7 + [ctypes.POINTER(ConvGeom), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_conv2d_dgrad_addmasked_bnred': (c_int, [c_void_p] * 9 + [ctypes.POINTER(ConvGeom), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_bwd_from_partials': (c_int, [c_void_p] * 8 + [c_int32, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int32, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_channel_scale_bwd_dm': (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int, c_void_p]),
    'mcn_bn_bwd_se': (c_int, [c_void_p] * 11 + [c_float, c_int32, c_int64, c_int32, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_bwd_se_sums': (c_int, [c_void_p] * 12 + [c_float, c_int32, c_int64, c_int32, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_act_scale_fwd': (c_int, [c_void_p] * 7 + [c_int32, c_int64, c_int32, c_int, c_int, c_void_p]),
    'mcn_se_bwd_sums_floats': (c_size_t, [c_int32, c_int64, c_int32, c_int]),
    'mcn_channel_scale_bwd_dm_bnsums': (c_int, [c_void_p] * 8 + [c_int32, c_int64, c_int32, c_int, c_void_p]),
    'mcn_bn_bwd_maxpool': (c_int, [c_void_p] * 10 + [c_float] + [c_int32] * 12 + [c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_fwd_train_fused_maxpool': (c_int, [c_void_p, c_void_p, c_int32, c_int32] + [c_void_p] * 10 + [c_float] + [c_int32] * 4 + [c_float] + [c_int32] * 8 + [c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_maxpool_fwd_affine_relu': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int32] * 12 + [c_int, c_void_p]),
    'mcn_bn_fwd_train_fused': (c_int, [c_void_p, c_void_p, c_int32, c_int32] + [c_void_p] * 11 + [c_float, c_int64, c_int32, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_fwd_infer': (c_int, [c_void_p] * 7 + [c_int64, c_int32, c_float, c_int, c_int, c_void_p]),
    'mcn_bn_bwd': (c_int, [c_void_p] * 12 + [c_float, c_int64, c_int32, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_bn_bwd_frozen': (c_int, [c_void_p] * 7 + [c_float] + [c_void_p] * 4 + [c_float, c_int64, c_int32, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_channel_affine': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    'mcn_dwconv2d_fwd': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int, c_void_p]),
    'mcn_dwconv2d_dgrad': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_int32, c_int, c_void_p]),
    'mcn_dwconv2d_workspace_bytes': (c_size_t, [ctypes.POINTER(ConvGeom), c_int]),
    'mcn_dwconv2d_wgrad': (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(ConvGeom), c_float, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_channel_repeat_fwd': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p]),
    'mcn_channel_repeat_bwd': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p]),
    'mcn_bias_grad_workspace_bytes': (c_size_t, [c_int64, c_int32]),
    'mcn_bias_grad': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_channel_scale_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int, c_void_p]),
    'mcn_channel_scale_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int, c_void_p]),
    'mcn_act_fwd': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'mcn_act_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'mcn_act_fwd_p': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    'mcn_act_bwd_p': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    'mcn_relu_fwd': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'mcn_relu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'mcn_add_relu_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'mcn_add_relu_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'mcn_accumulate': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    'mcn_cast': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    'mcn_input_prep': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_int, c_int, c_void_p]),
    'mcn_one_hot': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'mcn_maxpool_fwd': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int32] * 12 + [c_int, c_void_p]),
    'mcn_maxpool_bwd': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int32] * 12 + [c_int, c_void_p]),
    'mcn_avgpool_fwd': (c_int, [c_void_p, c_void_p] + [c_int32] * 12 + [c_int, c_void_p]),
    'mcn_avgpool_bwd': (c_int, [c_void_p, c_void_p] + [c_int32] * 12 + [c_int, c_void_p]),
    'mcn_global_avgpool_fwd': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int, c_void_p]),
    'mcn_global_avgpool_bwd': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int, c_void_p]),
    'mcn_global_avgpool_bwd_acc': (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int, c_void_p]),
    'mcn_fc_workspace_bytes': (c_size_t, [c_int32, c_int32, c_int32, c_int]),
    'mcn_fc_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_fc_bwd': (c_int, [c_void_p] * 6 + [c_float, c_int32, c_int32, c_int32, c_int, c_void_p, c_size_t, c_void_p]),
    'mcn_softmax_xent_fwd_bwd': (c_int, [c_void_p] * 8 + [c_int32, c_int32, c_float, c_float, c_void_p]),
    'mcn_l2_loss': (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mcn_decoupled_decay': (c_int, [c_void_p, c_int64, c_float, c_int32, c_float, c_void_p]),
    'mcn_sgd_nesterov_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64] + [c_float] * 6 + [c_void_p]),
    'mcn_sgd_nesterov_fused_h': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_int32, c_void_p]),
    'mcn_decoupled_decay_h': (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_float, c_void_p]),
    'mcn_ema_update_h': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'mcn_softmax_xent_rows_fwd_bwd': (c_int, [c_void_p] * 8 + [c_int64, c_int32, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    'mcn_softmax_xent_focal_fwd_bwd': (c_int, [c_void_p] * 8 + [c_int32, c_int32, c_float, c_float, c_float, c_float, c_void_p]),
    'mcn_softmax_xent_rows_focal_fwd_bwd': (c_int, [c_void_p] * 9 + [c_int64, c_int32, c_float, c_float, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    'mcn_l1_loss': (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mcn_l1_grad_h': (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    'mcn_softmax_xent_rows_soft_fwd_bwd': (c_int, [c_void_p] * 9 + [c_int64, c_int32, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    'mcn_resize_bilinear_fwd': (c_int, [c_void_p, c_void_p] + [c_int32] * 7 + [c_int, c_void_p]),
    'mcn_resize_bilinear_bwd': (c_int, [c_void_p, c_void_p] + [c_int32] * 7 + [c_int, c_void_p]),
    'mcn_copy_channels': (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_int64, c_int32, c_int, c_void_p]),
    'mcn_one_hot_seg': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    'mcn_clip_by_global_norm': (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mcn_clip_by_global_norm_runs': (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_int64), c_int32, c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mcn_ema_update': (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    'mcn_bn_running_chain': (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_float, c_void_p]),
    'mcn_bn_running_chain_strided': (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_float, c_void_p]),
}


def load(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError('libmcn_hip.so not found at {} — the HIP extension is required (no CPU fallback); '
                          'build it with `python myconvnet_amd/build.py`'.format(path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def _build_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_mcn_build', os.path.join(_HERE, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _ensure_fresh():
    """Before loading: the library must have been built from the sources of THIS tree (editing csrc/ or include/mcn.h and
    running tests / bench without rebuilding used to run the stale binary silently).  build.py records the source digest
    next to the library; on a mismatch (or a missing library) the library is rebuilt when hipcc is there, else the import
    fails with the build command.  MCN_SKIP_BUILD_CHECK=1 skips the check, MCN_NO_AUTOBUILD=1 turns the rebuild into an error."""
    if os.environ.get('MCN_SKIP_BUILD_CHECK') == '1' or os.environ.get('MCN_LIB_PATH'):
        return None
    mod = _build_module()
    want = mod.source_digest()
    have = None
    try:
        with open(LIB_PATH + '.id') as f:
            have = f.read().strip()
    except OSError:
        pass
    if have == want and os.path.exists(LIB_PATH):
        return want
    if os.environ.get('MCN_NO_AUTOBUILD') == '1' or not os.path.exists(mod.HIPCC):
        raise ImportError('libmcn_hip.so is {}: built from sources {}, the tree has {}... — rebuild with '
                          '`python myconvnet_amd/build.py`'.format('stale' if have else 'missing', (have or 'none')[:12], want[:12]))
    # N ranks of one node import this module at the same time (bench.py under torch.distributed.run): ONE of them rebuilds, the others
    # wait on the lock and find the fresh library when they get it (before any GPU call: nothing here touches the device)
    import fcntl
    with open(LIB_PATH + '.lock', 'w') as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            try:
                with open(LIB_PATH + '.id') as f:
                    have = f.read().strip()
            except OSError:
                have = None
            if not (have == want and os.path.exists(LIB_PATH)):
                mod.build(verbose=False)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return want


_want = _ensure_fresh()
lib = load()
if _want is not None and lib.mcn_build_id().decode() != _want:
    raise ImportError('libmcn_hip.so reports build id {}..., expected {}... — rebuild with `python myconvnet_amd/build.py --force`'
                      .format(lib.mcn_build_id().decode()[:12], _want[:12]))


# the library named by MCN_LIB_PATH is the CPU build: tensors live on the host, the stream argument is ignored
IS_CPU_LIB = lib.mcn_build_id().decode() == 'cpu'


def last_error():
    return lib.mcn_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != 0:
        raise McnError(rc, last_error())


def conv_geom(N, H, W, Cin, Cout, KH, KW, SH, SW, DH, DW, pads, x_cs=0, tile=0):
    pt, pb, pl, pr = pads
    return ConvGeom(N, H, W, Cin, Cout, KH, KW, SH, SW, DH, DW, pt, pb, pl, pr, x_cs, tile)
