"""ctypes loader of libpt_hip.so (the C-ABI of include/pt_hip.h).

There is no CPU fallback: if the library is missing or cannot be loaded the
import of anything that renders fails with the reason.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_hip.so")

# Every symbol include/pt_hip.h declares: (name, restype, argtypes)
_VP, _I, _SZ = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = [
    ("pt_create", _I, [_I, _I, _I, C.POINTER(_VP)]),
    ("pt_destroy", None, [_VP]),
    ("pt_update_scene", _I, [_VP, _VP, _I, _VP, _I, _VP, _I, _VP, _I, _VP, _I, _VP, _SZ]),
    ("pt_reset_voxels", _I, [_VP]),
    ("pt_render", _I, [_VP, _VP]),
    ("pt_present", _I, [_VP]),
    ("pt_readback_rgba8", _I, [_VP, _VP, _SZ]),
    ("pt_readback_f32", _I, [_VP, _VP, _SZ]),
    ("pt_shader_count", _I, []),
    ("pt_shader_path", C.c_char_p, [_I]),
    ("pt_shader_index", _I, [C.c_char_p]),
    ("pt_last_error", C.c_char_p, []),
    ("pt_set_row_bands", _I, [_VP, _I, _I, _I]),
    ("pt_local_rows", _I, [_VP]),
    ("pt_assemble_bands", _I, [_VP, _VP, _VP, _I, _I, _I, _I]),
    ("pt_comm_unique_id", _I, [_VP, _SZ]),
    ("pt_comm_create", _I, [_VP, _SZ, _I, _I, _I, C.POINTER(_VP)]),
    ("pt_comm_destroy", None, [_VP]),
    ("pt_gather_bands", _I, [_VP, _VP, _I, _VP]),
    ("pt_device_assembled", _VP, [_VP]),
    ("pt_readback_assembled", _I, [_VP, _VP, _SZ]),
    ("pt_stream_concurrency", _I, [_VP, C.POINTER(_I)]),
    ("pt_wave_timeline", _I, [_VP, _VP, _VP, _SZ, C.POINTER(_I)]),
    ("pt_render_counted", _I, [_VP, _VP, _VP]),
    ("pt_render_samples", _I, [_VP, _VP, _I, _I, _I]),
    ("pt_render_static", _I, [_VP, _VP, _VP, _VP, _VP]),
    ("pt_set_slot", _I, [_VP, _I, _VP, _VP, _VP]),
    ("pt_last_slot", _I, [_VP]),
    ("pt_device_rgba8", _VP, [_VP]),
    ("pt_device_f32", _VP, [_VP]),
    ("pt_set_output", _I, [_VP, _VP, _VP]),
    ("pt_set_stream", _I, [_VP, _VP]),
    ("pt_timer_begin", _I, [_VP]),
    ("pt_timer_end", _I, [_VP]),
    ("pt_timer_elapsed_ms", _I, [_VP, C.POINTER(C.c_float)]),
    ("pt_set_option", _I, [_VP, _I, _I]),
    ("pt_device_info", _I, [_VP, C.c_char_p, _SZ, C.POINTER(_I)]),
    ("pt_last_launch", _I, [_VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
]

_lib = None


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pt_hip error {code}: {message}")
        self.code = code


def load() -> C.CDLL:
    """Load libpt_hip.so and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback"
        )
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise PtError(rc, load().pt_last_error().decode("utf-8", "replace"))
