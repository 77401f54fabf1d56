"""ctypes binding of libpvr_hip.so (include/pvr_hip.h).  No CPU fallback: if the library is
missing the import of the product path fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PVR_LIB') or os.path.join(_HERE, 'lib', 'libpvr_hip.so')      # PVR_LIB: e.g. the host-ASan build (`make asan`)

PVR_BF16, PVR_F16, PVR_F32 = 0, 1, 2
ARCH_RESNET50, ARCH_RESNET50_L4, ARCH_RESNET50_L3 = 0, 1, 2


class EncoderDesc(C.Structure):
    _fields_ = [('arch', C.c_int32), ('dtype', C.c_int32), ('max_batch', C.c_int32), ('chunk', C.c_int32),
                ('resize', C.c_int32), ('crop', C.c_int32), ('mean', C.c_float * 3), ('std_', C.c_float * 3)]


_lib = None

_SIGS = {
    'pvr_version': (C.c_char_p, []),
    'pvr_has_experiments': (C.c_int32, []),
    'pvr_debug_stem_u8_geometry_ok': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'pvr_encoder_set_host_backend': (C.c_int, [C.c_void_p, C.c_int32]),
    'pvr_stage_copy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32]),
    'pvr_last_error': (C.c_size_t, [C.c_char_p, C.c_size_t]),
    'pvr_encoder_create': (C.c_int, [C.POINTER(EncoderDesc), C.POINTER(C.c_void_p)]),
    'pvr_encoder_load_weights': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    'pvr_encoder_finalize': (C.c_int, [C.c_void_p]),
    'pvr_encoder_out_size': (C.c_int32, [C.c_void_p]),
    'pvr_encoder_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'pvr_encoder_forward_lane': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    'pvr_debug_set_conv_algo': (C.c_int, [C.c_int32]),
    'pvr_file_sizes': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    'pvr_read_files': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]),
    'pvr_png_scratch_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    'pvr_png_decode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'pvr_debug_conv_expand_launches': (C.c_int64, []),
    'pvr_debug_pp_persistent_launches': (C.c_int64, []),
    'pvr_encoder_set_crop_position': (C.c_int, [C.c_void_p, C.c_int32]),
    'pvr_encoder_set_low_latency': (C.c_int, [C.c_void_p, C.c_int32]),
    'pvr_encoder_tap': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    'pvr_encoder_debug_stop_after': (C.c_int, [C.c_void_p, C.c_char_p]),
    'pvr_encoder_debug_set_fusion': (C.c_int, [C.c_void_p, C.c_int32]),
    'pvr_encoder_launch_name': (C.c_int32, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]),
    'pvr_encoder_destroy': (None, [C.c_void_p]),
    'pvr_encoder_profile': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    'pvr_encoder_profile_span': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                           C.POINTER(C.c_float)]),
    'pvr_op_preprocess': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    'pvr_op_stem': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'pvr_op_maxpool': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'pvr_op_conv2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 12 + [C.c_void_p]),
    'pvr_op_avgpool': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'pvr_debug_convert': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    'pvr_op_bneck_frame': (C.c_int, [C.c_void_p] * 13 + [C.c_int32] * 3 + [C.c_void_p]),
    'pvr_debug_bneck_frame_launches': (C.c_int64, []),
    'pvr_debug_set_frame64': (C.c_int, [C.c_int32]),
    'pvr_debug_bneck_frame64_launches': (C.c_int64, []),
    'pvr_debug_bneck_frame64_stamps': (C.c_int, [C.c_void_p] * 8 + [C.c_int32] * 2 + [C.c_void_p] * 2),
    'pvr_op_conv_wfrag': (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 12 + [C.c_void_p]),
    'pvr_debug_conv_wfrag_launches': (C.c_int64, []),
    'pvr_op_conv_wfrag_pool': (C.c_int, [C.c_void_p] * 5 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    'pvr_op_conv2d_dual': (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 15 + [C.c_void_p]),
    'pvr_op_pack_frag_weights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'pvr_debug_bneck_frame_stamps': (C.c_int, [C.c_void_p] * 12 + [C.c_int32] * 2 + [C.c_void_p] * 2),
    'pvr_encoder_check_range': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_int32)]),
    'pvr_encoder_debug_set_switch': (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    'pvr_encoder_launch_kernel': (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]),
    'pvr_op_split16_pack_weights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'pvr_op_conv2d_split16': (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 9 + [C.c_void_p]),
    'pvr_op_conv2d_f32': (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 9 + [C.c_void_p]),
    'pvr_debug_conv_split16_launches': (C.c_int64, []),
    'pvr_debug_set_stem_regpool': (C.c_int, [C.c_int32]),
    'pvr_debug_chain_wave128_launches': (C.c_int64, []),
    'pvr_op_nonfinite_flag': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
}


def lib():
    """The loaded library (raises if it has not been built: run `python __graft_entry__.py build`)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError('libpvr_hip.so not found at %s: build it with `make -C pvr_habitat_amd/csrc` '
                               '(there is no CPU fallback for the HIP path)' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if not hasattr(l, name):
                continue                      # optional groups (policy) are bound by their own modules
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    lib().pvr_last_error(buf, 1024)
    return buf.value.decode(errors='replace')


def check(status):
    if status != 0:
        raise RuntimeError('libpvr_hip: %s (status %d)' % (last_error(), status))


def stream_ptr():
    """Raw hipStream_t of torch's current stream."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError('pvr_habitat_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU path')
