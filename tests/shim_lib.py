"""ctypes binding of the per-frame drop-in library (libmbe_neo_amd.so, include/mbe_neo_amd.h)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "mbelib-neo_amd", "libmbe_neo_amd.so")
HEADER = os.path.join(ROOT, "include", "mbe_neo_amd.h")

_vp = C.c_void_p


def declared_symbols():
    return sorted(set(re.findall(r"\b(mbe_[A-Za-z0-9_]+)\s*\(", open(HEADER).read())))


def load():
    try:
        import torch  # noqa: F401  (same HIP runtime instance as torch, see mbelib-neo_amd/_native.py)
    except ImportError:
        pass
    h = C.CDLL(PATH)
    for name in declared_symbols():
        fn = getattr(h, name)
        fn.argtypes = None
    for name in ("mbe_processImbe7200x4400Framef", "mbe_processImbe7200x4400Frame", "mbe_processAmbe3600x2450Framef",
                 "mbe_processAmbe3600x2450Frame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 7
    for name in ("mbe_processImbe4400Dataf", "mbe_processImbe4400Data", "mbe_processAmbe2450Dataf", "mbe_processAmbe2450Data"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 6
    for name in ("mbe_decodeImbe7200x4400Frame", "mbe_decodeAmbe3600x2450Frame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 3
    for name in ("mbe_golay2312", "mbe_hamming1511"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp, _vp]
    for name in ("mbe_processImbe7200x4400SoftFramef", "mbe_processImbe7200x4400SoftFrame",
                 "mbe_processAmbe3600x2450SoftFramef", "mbe_processAmbe3600x2450SoftFrame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 7
    for name in ("mbe_decodeImbe7200x4400SoftFrame", "mbe_decodeAmbe3600x2450SoftFrame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 3
    for name in ("mbe_processImbe7100x4400Framef", "mbe_processImbe7100x4400Frame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 7
    h.mbe_decodeImbe7100x4400Frame.restype = C.c_int
    h.mbe_decodeImbe7100x4400Frame.argtypes = [_vp] * 3
    h.mbe_7100x4400hamming1511.restype = C.c_int
    h.mbe_7100x4400hamming1511.argtypes = [_vp, _vp]
    for name in ("mbe_processAmbe3600x2400Framef", "mbe_processAmbe3600x2400Frame", "mbe_processAmbe3600x2400SoftFramef",
                 "mbe_processAmbe3600x2400SoftFrame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 7
    for name in ("mbe_processAmbe2400Dataf", "mbe_processAmbe2400Data"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 6
    for name in ("mbe_decodeAmbe3600x2400Frame", "mbe_decodeAmbe3600x2400SoftFrame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 3
    for name in ("mbe_processImbe7100x4400SoftFramef", "mbe_processImbe7100x4400SoftFrame"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp] * 7
    h.mbe_decodeImbe7100x4400SoftFrame.restype = C.c_int
    h.mbe_decodeImbe7100x4400SoftFrame.argtypes = [_vp] * 3
    for name in ("mbe_golay2312Soft", "mbe_hamming1511Soft", "mbe_7100x4400hamming1511Soft"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp, _vp]
    h.mbe_softBitsFromLlr.restype = C.c_int
    h.mbe_softBitsFromLlr.argtypes = [_vp, _vp, C.c_size_t]
    h.mbe_softBitsFromHard.restype = C.c_int
    h.mbe_softBitsFromHard.argtypes = [_vp, _vp, C.c_size_t, C.c_uint8]
    h.mbe_checkGolayBlock.restype = C.c_int
    h.mbe_checkGolayBlock.argtypes = [C.POINTER(C.c_long)]
    h.mbe_setThreadRngSeed.argtypes = [C.c_uint32]
    h.mbe_setThreadRngSeed.restype = None
    for name, n in (("mbe_initMbeParms", 3), ("mbe_moveMbeParms", 2), ("mbe_useLastMbeParms", 2), ("mbe_synthesizeSpeechf", 3),
                    ("mbe_synthesizeSpeech", 3), ("mbe_floattoshort", 2), ("mbe_spectralAmpEnhance", 1),
                    ("mbe_applyAdaptiveSmoothing", 2), ("mbe_synthesizeComfortNoisef", 1), ("mbe_synthesizeComfortNoise", 1),
                    ("mbe_synthesizeSilencef", 1), ("mbe_synthesizeSilence", 1), ("mbe_initProcessResult", 1)):
        getattr(h, name).restype = None
        getattr(h, name).argtypes = [_vp] * n
    for name in ("mbe_requiresMuting", "mbe_isMaxFrameRepeat", "mbe_requiresAdaptiveSmoothing"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp]
    h.mbe_synthesizeTonef.restype = None
    h.mbe_synthesizeTonef.argtypes = [_vp, _vp, _vp]
    h.mbe_synthesizeTonefdstar.restype = None
    h.mbe_synthesizeTonefdstar.argtypes = [_vp, _vp, _vp, C.c_int]
    h.mbe_formatProcessResult.restype = None
    h.mbe_formatProcessResult.argtypes = [C.c_char_p, C.c_size_t, _vp]
    h.mbe_versionString.restype = C.c_char_p
    for name in ("mbe_eccImbe7200x4400C0", "mbe_demodulateImbe7200x4400Data", "mbe_eccAmbe3600x2450C0", "mbe_demodulateAmbe3600x2450Data",
                 "mbe_eccAmbe3600x2400C0", "mbe_demodulateAmbe3600x2400Data", "mbe_eccImbe7100x4400C0",
                 "mbe_demodulateImbe7100x4400Data", "mbe_convertImbe7100to7200"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp]
    for name in ("mbe_eccImbe7200x4400Data", "mbe_eccAmbe3600x2450Data", "mbe_eccAmbe3600x2400Data", "mbe_eccImbe7100x4400Data"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp, _vp]
    for name in ("mbe_decodeImbe4400Parms", "mbe_decodeAmbe2450Parms", "mbe_decodeAmbe2400Parms"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = [_vp, _vp, _vp]
    for name in ("mbe_dumpAmbe2400Data", "mbe_dumpAmbe3600x2400Frame", "mbe_dumpAmbe2450Data", "mbe_dumpAmbe3600x2450Frame",
                 "mbe_dumpImbe4400Data", "mbe_dumpImbe7200x4400Data", "mbe_dumpImbe7200x4400Frame", "mbe_dumpImbe7100x4400Data",
                 "mbe_dumpImbe7100x4400Frame"):
        getattr(h, name).restype = None
        getattr(h, name).argtypes = [_vp]
    h.mbe_batchBegin.restype = C.c_int
    h.mbe_batchBegin.argtypes = [C.c_int]
    for name in ("mbe_flush", "mbe_batchPending", "mbe_batchEnd"):
        getattr(h, name).restype = C.c_int
        getattr(h, name).argtypes = []
    h.mbe_batchRelease.restype = C.c_int
    h.mbe_batchRelease.argtypes = [_vp]
    return h


def p(a):
    return a.ctypes.data if a is not None else None
