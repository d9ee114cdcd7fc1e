"""mbelib-neo_amd -- MI355X-native batched MBE (IMBE 7200x4400 / AMBE+2 3600x2450) decode path.

Only the hot path lives here: a C-ABI HIP launcher (``libmbx_hip.so``, sources in ``csrc/``)
and the thin host layer over it.  There is no CPU compute path in this package: every
function that produces decoded parameters or PCM requires the HIP library and a GPU and
raises ``NativeLibraryError`` otherwise.
"""
from .layout import (  # noqa: F401
    PARMS_DTYPE,
    RESULT_DTYPE,
    RNG_DTYPE,
    RECORD_DTYPE,
    CODEC_IMBE7200X4400,
    CODEC_AMBE3600X2450,
    FRAME_BYTES,
    PARAM_BITS,
    FLAG_C0_VALID,
    FLAG_C4_VALID,
    FLAG_TONE,
    FLAG_ERASURE,
    FLAG_REPEAT,
    FLAG_MUTE,
    STATUS_INVALID_ARGUMENT,
    STATUS_INVALID_BITS,
    init_parms,
    init_state,
    rng_default,
    rng_seeded,
    load_tables_blob,
)
from ._native import NativeLibraryError, lib, library_path  # noqa: F401
from . import framegen  # noqa: F401
from .decoder import BatchDecoder  # noqa: F401

__all__ = ["BatchDecoder", "framegen", "lib", "NativeLibraryError"]
