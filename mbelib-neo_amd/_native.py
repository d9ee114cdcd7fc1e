"""ctypes binding of the C-ABI HIP launcher (include/mbx.h).  Fails loudly when the library
is missing or cannot be initialised: there is no Python/CPU stand-in."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libmbx_hip.so"


class NativeLibraryError(RuntimeError):
    pass


def library_path():
    """The product library, or the build named by MBX_HIP_LIBRARY: for tools/ the development builds (libmbx_hip_ablate.so, which
    additionally exports mbx_debug_set_ablation), for one test the fault-injection build (libmbx_hip_testing.so: mbx_testing_set_front_skip)."""
    return os.environ.get("MBX_HIP_LIBRARY") or os.path.join(_HERE, _LIB_NAME)


_lib = None

_vp = C.c_void_p
_sz = C.c_size_t
_SIGNATURES = {
    "mbx_init": (C.c_int, [C.c_int, _vp, _sz]),
    "mbx_shutdown": (None, []),
    "mbx_reserve": (C.c_int, [_sz]),
    "mbx_reserve_stream": (C.c_int, [_vp, _sz]),
    "mbx_uses_expand_launch": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mbx_release_stream": (C.c_int, [_vp]),
    "mbx_workspace_bytes": (_sz, [_sz]),
    "mbx_device_ready": (C.c_int, [C.c_int]),
    "mbx_set_stream_order": (C.c_int, [C.c_int]),
    "mbx_set_tone_synthesis": (C.c_int, [C.c_int]),
    "mbx_table_checksum": (C.c_uint32, []),
    "mbx_last_error": (C.c_char_p, []),
    "mbx_comm_unique_id": (C.c_int, [_vp]),
    "mbx_comm_init": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int]),
    "mbx_comm_destroy": (C.c_int, [_vp]),
    "mbx_comm_agree": (C.c_int, [_vp, C.c_uint32, _vp, _vp]),
    "mbx_init_broadcast": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _sz, _vp, _vp]),
    "mbx_pack_imbe7200x4400": (C.c_int, [_vp, _sz, _vp]),
    "mbx_pack_ambe3600x2450": (C.c_int, [_vp, _sz, _vp]),
    "mbx_pack_cells": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "mbx_wire_bit_of_cell": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "mbx_wire_permutation": (C.c_int, [C.c_int, _vp, _vp, C.c_int, _vp]),
    "mbx_unpack_records": (None, [_vp, _sz, C.c_int, _vp, _vp]),
    "mbx_fec_imbe7200x4400": (C.c_int, [_vp, _sz, _vp, _vp]),
    "mbx_fec_ambe3600x2450": (C.c_int, [_vp, _sz, _vp, _vp]),
    "mbx_process_records": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_process_records_ws": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mbx_process_batch_ws": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mbx_expand_records": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "mbx_stream_expanded": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_process_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_stage_in": (C.c_int, [_vp, _vp, _sz, _vp]),
    "mbx_frame_server_start": (C.c_int, [_vp, C.c_uint, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_process_frame": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp]),
    "mbx_process_frame_shadow": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, C.c_int, _vp, _vp]),
    "mbx_process_batch_indexed": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_process_batch_resident": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_resident_materialize": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp]),
    "mbx_expand_records_ws": (C.c_int, [C.c_int, _vp, _sz, _vp, _sz, _vp]),
    "mbx_stream_expanded_ws": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mbx_stream_expanded_resident": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_fec_stage": (C.c_int, [C.c_int, C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "mbx_decode_parms": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp, _vp, _vp]),
    "mbx_synthesize_speech": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_floattoshort": (C.c_int, [_vp, _vp, _sz, _vp]),
    "mbx_result_histogram": (C.c_int, [_vp, _sz, _vp, _vp]),
    "mbx_spectral_amp_enhance": (C.c_int, [C.c_int, _vp, _vp]),
    "mbx_adaptive_smoothing": (C.c_int, [C.c_int, _vp, _vp, _vp]),
    "mbx_comfort_noise": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp]),
    "mbx_ecc_words": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "mbx_synthesize_tone": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_state_copy": (C.c_int, [C.c_int, _vp, _vp]),
    "mbx_process_batch_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_synthesize_speech_host": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "mbx_floattoshort_host": (C.c_int, [_vp, _vp, _sz]),
    "mbx_fec_host": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "mbx_pack_imbe7100x4400": (C.c_int, [_vp, _sz, _vp]),
    "mbx_fec_imbe7100x4400": (C.c_int, [_vp, _sz, _vp, _vp]),
    "mbx_fec_soft": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp]),
    "mbx_process_batch_soft": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_ecc_soft_words": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp, _vp]),
    "mbx_validate_soft_bits": (C.c_int, [_vp, _sz]),
    "mbx_soft_bits_from_hard": (C.c_int, [_vp, _vp, _sz, C.c_uint8]),
    "mbx_soft_bits_from_llr": (C.c_int, [_vp, _vp, _sz]),
    "mbx_fec_soft_host": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "mbx_process_batch_soft_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_ecc_soft_words_host": (C.c_int, [C.c_int, _vp, _sz, _vp, _vp]),
    "mbx_session_create": (C.c_int, [_vp, C.c_int, C.c_int, _sz, C.c_uint]),
    "mbx_session_destroy": (C.c_int, [_vp]),
    "mbx_session_streams": (C.c_int, [_vp]),
    "mbx_session_submit": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "mbx_session_submit_indexed": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mbx_session_wait": (C.c_int, [_vp]),
    "mbx_session_reset": (C.c_int, [_vp, C.c_int, C.c_int]),
    "mbx_session_seed": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "mbx_session_get_state": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "mbx_session_set_state": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "mbx_host_alloc": (_vp, [_sz]),
    "mbx_host_free": (None, [_vp]),
    "mbx_rng_default": (None, [_vp]),
    "mbx_rng_seed": (None, [_vp, C.c_uint32]),
    "mbx_stream_kernel_name": (C.c_char_p, [C.c_int, C.c_int]),
    "mbx_batch_kernel_name": (C.c_char_p, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mbx_front_fallbacks": (C.c_longlong, [C.c_void_p]),
    "mbx_launch_slices": (C.c_int, [C.c_int, C.c_int, C.c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib():
    """The loaded library handle (loads on first use)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise NativeLibraryError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  mbelib-neo_amd has no CPU fallback."
            )
        # torch ships its own HIP runtime; load it first so that this library binds to the SAME
        # libamdhip64 instance (device pointers and streams are shared with torch tensors).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            handle = C.CDLL(path)
        except OSError as e:  # e.g. libamdhip64 not found
            raise NativeLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                if os.environ.get("MBX_HIP_LIBRARY_ALLOW_OLDER") == "1":   # tools/abx.sh: A/B against a library of an earlier commit
                    continue
                raise NativeLibraryError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if hasattr(handle, "mbx_testing_set_front_skip"):   # libmbx_hip_testing.so only (tests/front_skip_case.py)
            handle.mbx_testing_set_front_skip.restype = C.c_int
            handle.mbx_testing_set_front_skip.argtypes = [C.c_int]
        if hasattr(handle, "mbx_debug_set_ablation"):   # development build only
            handle.mbx_debug_set_ablation.restype = None
            handle.mbx_debug_set_ablation.argtypes = [C.c_int]
        _lib = handle
    return _lib


def check(rc, what):
    if rc < 0:
        msg = lib().mbx_last_error()
        raise NativeLibraryError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
    return rc
