"""ctypes binding of liblass_hip.so (include/lass_hip.h).  No CPU fallback: a missing library is a hard error."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LASS_HIP_LIB") or os.path.join(_HERE, "csrc", "liblass_hip.so")  # env: diagnostic builds

_lib = None

# every symbol include/lass_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("lass_version", c_int, []),
    ("lass_create", c_int, [POINTER(c_void_p), c_int]),
    ("lass_create_multistft", c_int, [POINTER(c_void_p), c_int, c_int, c_int, POINTER(c_int), c_int]),
    ("lass_destroy", c_int, [c_void_p]),
    ("lass_last_error", c_char_p, [c_void_p]),
    ("lass_set_param", c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int, c_int]),
    ("lass_finalize", c_int, [c_void_p, c_int]),
    ("lass_workspace_bytes", c_int, [c_void_p, c_int, c_int, POINTER(c_size_t)]),
    ("lass_separate", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    ("lass_stft_magphase", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    ("lass_mix_at_snr", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("lass_segment_mix", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    ("lass_multi_stft", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_void_p),
                                POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    ("lass_graph_stats", c_int, [c_void_p, POINTER(c_long), POINTER(c_long)]),
    ("lass_set_graph_replay", c_int, [c_void_p, c_int]),
    ("lass_separate_components", c_int, [c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_int, c_void_p, c_size_t, c_void_p]),
    ("lass_stft_components", c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int),
                                     POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    ("lass_istft_nfft", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    ("lass_istft", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("lass_film", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    ("lass_film_width", c_int, [c_void_p]),
    ("lass_film_offset", c_int, [c_void_p, c_char_p]),
    ("lass_film_raw", c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    ("lass_convblock", c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    ("lass_encoder_block", c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    ("lass_front_end", c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("lass_workspace_tensor", c_int, [c_void_p, c_int, c_int, c_char_p, POINTER(c_size_t), POINTER(c_int64),
                                      POINTER(c_int64)]),
    ("lass_upconv", c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    ("lass_mask_apply", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                c_void_p, c_void_p]),
    ("lass_sdr_stats", c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    ("lass_set_profiling", c_int, [c_void_p, c_int]),
    ("lass_profile_count", c_int, [c_void_p]),
    ("lass_profile_get", c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_double), POINTER(c_int)]),
    ("lass_profile_reset", c_int, [c_void_p]),
]


class LassError(RuntimeError):
    pass


def load():
    """Load liblass_hip.so (once).  torch is imported first so the library binds to the HIP runtime torch loaded."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must precede the dlopen: one libamdhip64 per process)

    if not os.path.exists(LIB_PATH):
        raise LassError(
            f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  lass_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc: int, what: str):
    if rc < 0:
        msg = load().lass_last_error(ctx)
        raise LassError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
    return rc
