"""ctypes binding of libgenie_hip.so (include/genie_hip.h).

The library is the product path: importing this module builds nothing and
falls back to nothing -- if the shared object is missing or a symbol is absent
it raises, loudly.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime of the
# process (streams and device pointers are shared with torch), so it has to be
# loaded before libgenie_hip.so resolves the same SONAME.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GENIE_HIP_LIB', os.path.join(_HERE, 'lib', 'libgenie_hip.so'))


class GenieDims(C.Structure):
    """genie_dims_t -- field order is the ABI."""
    _fields_ = [
        ('c_s', C.c_int32), ('c_p', C.c_int32),
        ('c_pos_emb', C.c_int32), ('c_chain_emb', C.c_int32), ('c_timestep_emb', C.c_int32),
        ('relpos_k', C.c_int32),
        ('template_dist_n_bin', C.c_int32),
        ('template_dist_min', C.c_float), ('template_dist_step', C.c_float),
        ('n_pair_transform_layer', C.c_int32), ('c_hidden_mul', C.c_int32), ('pair_transition_n', C.c_int32),
        ('n_structure_layer', C.c_int32), ('n_structure_block', C.c_int32),
        ('c_hidden_ipa', C.c_int32), ('n_head_ipa', C.c_int32), ('n_qk_point', C.c_int32), ('n_v_point', C.c_int32),
        ('rescale', C.c_float),
        ('n_timestep', C.c_int32), ('max_n_res', C.c_int32), ('max_n_chain', C.c_int32),
    ]


class GenieFeatures(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        'aatype', 'atom_positions', 'residue_mask', 'residue_index', 'chain_index',
        'fixed_sequence_mask', 'fixed_structure_mask', 'interface_mask')]


class GenieTaps(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ('s', 'p', 's_final', 'rots_out', 'trans_out', 'p_init', 'p_layer0', 'states', 'p_trimul_out0',
                                              'ipa_cat0')]


class GenieGemmDesc(C.Structure):
    """genie_gemm_desc_t (include/genie_hip.h): one strided, batched GEMM of the training path."""
    _fields_ = [(n, C.c_int32) for n in ('M', 'N', 'K', 'batch', 'nb2', 'nsplit', 'mode', 'terms', 'relu')] + \
               [(n, C.c_int64) for n in ('am', 'ak', 'bk', 'bn', 'cm', 'cn', 'a1', 'a2', 'b1', 'b2', 'c1', 'c2')] + [('alpha', C.c_float)] + \
               [('cblk', C.c_int32), ('cblk_m', C.c_int32), ('ctab', C.c_int64 * 8), ('atab', C.c_int64 * 8)]


class GenieTrainOpts(C.Structure):
    _fields_ = [('tri_dropout', C.c_float), ('ipa_dropout', C.c_float), ('transition_dropout', C.c_float), ('seed', C.c_uint32),
                ('train_mode', C.c_int32), ('fast_math', C.c_int32), ('struct_done_event', C.c_void_p)]


# name -> (restype, argtypes); every symbol include/genie_hip.h declares
SYMBOLS = {
    'genie_create': (C.c_int, [C.POINTER(GenieDims), C.c_int, C.POINTER(C.c_void_p)]),
    'genie_destroy': (None, [C.c_void_p]),
    'genie_last_error': (C.c_char_p, [C.c_void_p]),
    'genie_weight_count': (C.c_size_t, [C.POINTER(GenieDims)]),
    'genie_load_weights': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'genie_set_tables': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'genie_prepare_features': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(GenieFeatures)]),
    'genie_frenet': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'genie_frenet_frames': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'genie_denoise': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.POINTER(GenieTaps)]),
    'genie_q_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'genie_training_loss': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    'genie_adam_step': (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_int]),
    'genie_train_forward_backward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_float, C.POINTER(GenieTrainOpts), C.c_void_p, C.c_void_p]),
    'genie_denoise_vjp': (C.c_int, [C.c_void_p] * 10),
    'genie_train_workspace_bytes': (C.c_size_t, [C.c_void_p]),
    'genie_train_kept_bytes': (C.c_size_t, [C.c_void_p]),
    'genie_train_gemm_flop': (C.c_double, [C.c_void_p]),
    'genie_train_gemm': (C.c_int, [C.c_void_p, C.POINTER(GenieGemmDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'genie_p_sample': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    'genie_sample_loop': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    'genie_set_math': (C.c_int, [C.c_void_p, C.c_int]),
    'genie_get_math': (C.c_int, [C.c_void_p]),
    'genie_profile_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'genie_profile_read': (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int64), C.c_int]),
    'genie_probe_mfma': (C.c_int, [C.c_void_p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'genie_workspace_bytes': (C.c_size_t, [C.c_void_p]),
}

_lib = None


class GenieError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree library and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GenieError(
            f'{LIB_PATH} not found: build it with `python -m genie2_amd.build` '
            '(there is no CPU fallback for the denoising path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(handle, rc, what):
    if rc != 0:
        msg = load_library().genie_last_error(handle)
        raise GenieError(f'{what} failed (rc={rc}): {msg.decode() if msg else "?"}')
