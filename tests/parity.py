"""Comparison helpers shared by the oracle-vs-golden and HIP-vs-oracle tests.

Tolerances: everything integer is bit-exact (parameter bits, error counts,
flags, return codes, L/K/Vl, repeat counters, thresholds, the LCG noise state); float PCM
relative RMS <= 1e-4 per batch and <= 1e-3 for the worst single frame; int16 PCM within
1 LSB on >= 99.9 % of samples and never more than 3 LSB -- except in frames that are driven INTO THE SOFT CLIP
(a float sample of the reference at +-4446.95, i.e. harmonic amplitudes beyond the output range: random channel bits
decode to 60,000-80,000 against +-4,447), where the bound is 4 LSB: the worst difference observed in 94 M samples, pinned
by the fixture below (SURVEY.md section 8(c) says 3 everywhere; this is the one stated deviation from it).  check_pcm also
fails when more than 35 % of a workload's frames fall under the clipped bound (random channel bits: 20 %), so a workload
cannot drift under the looser bound unnoticed.

WHICH CHECKER: `ref_*` arguments come either from the ORACLE (oracle/mbx_oracle.c: the CPU restatement.  Its unvoiced FFT is
double precision by default -- 6e-8 relative away from the reference's float PFFFT, every integer and every decision identical
to the real reference on 131,072 random frames per codec, oracle/_ref/cmp_ref -- and, after set_fft_float(1), FFTPACK's float
transform as PFFFT runs it: then the oracle is the reference bit for bit, tests/test_oracle_golden.py) or from a GOLDEN FIXTURE
written by the real reference (tests/golden/*.bin via oracle/_ref).  Each test says in its docstring which it uses.

Why the clipped frames have their own bound (tests/golden/tail_cases.npz, test_tail_cases_*): the samples that are NOT
clipped in such a frame are where a sum of amplitude ~1e5 happens to cross the output range, so an error of 5e-6 of the
amplitude is 3 LSB.  Measured over 94 M samples of random-bit frames (tools/find_tail.py, three seeds x four codecs): the
HIP path is within 1 LSB of the oracle on 99.9998 %, differs by 3 LSB on three samples and by 4 LSB on one, all in
clipped frames; on those very frames the REFERENCE's own build for an FMA target (its -std=gnu99 defaults,
oracle/Makefile `fma`) differs from its IEEE build by 8 to 55 LSB, and over 10.5 M samples by >= 7 LSB on 4,000 of them
(oracle/tools/ref_simd_vs_scalar.py fma).  The oracle (and the HIP path) follow the IEEE build.
"""
import numpy as np

from mbelib_neo_amd.layout import EXACT_FLOAT_FIELDS, FLOAT_FIELDS, INT_FIELDS

PCM_REL_RMS = 1e-4
PCM_WORST_FRAME = 1e-3
STATE_REL_RMS = 1e-4
INT16_MAX_LSB = 3            # frames below the clip
INT16_MAX_LSB_CLIPPED = 4    # frames with a sample at the soft-clip level: the observed worst case (see the module docstring)
MAX_CLIPPED_SHARE = 0.35     # of a workload's frames (random channel bits drive 20 % of the frames into the clip)
CLIP_LEVEL = 32767.0 * 0.95 / 7.0   # ref src/core/mbelib.c:1148-1177 soft clip of the float PCM


def clipped_frames(ref_f):
    """frames of float PCM [n, 160] in which the reference reaches the soft-clip level"""
    return np.abs(np.asarray(ref_f).reshape(-1, 160)).max(axis=1) >= CLIP_LEVEL - 0.01


def rel_rms(ref, got):
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64)
    ok = np.isfinite(ref) & np.isfinite(got)
    assert np.array_equal(np.isnan(ref), np.isnan(got)), "NaN pattern differs"
    den = np.sum(ref[ok] ** 2)
    num = np.sum((ref[ok] - got[ok]) ** 2)
    return float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))


def check_pcm(ref_f, got_f, ref_s=None, got_s=None, rel=PCM_REL_RMS, worst=PCM_WORST_FRAME, what="pcm"):
    ref_f = np.asarray(ref_f).reshape(-1, 160)
    got_f = np.asarray(got_f).reshape(-1, 160)
    total = rel_rms(ref_f, got_f)
    assert total <= rel, f"{what}: relative RMS {total:.3e} > {rel:.1e}"
    # worst frame, relative to the batch RMS level so silent frames do not blow the ratio up
    level = np.sqrt(np.mean(ref_f.astype(np.float64) ** 2)) + 1e-30
    err = np.sqrt(np.mean((ref_f.astype(np.float64) - got_f.astype(np.float64)) ** 2, axis=1))
    frame_rms = np.sqrt(np.mean(ref_f.astype(np.float64) ** 2, axis=1))
    ratio = err / np.maximum(frame_rms, 0.05 * level)
    assert ratio.max() <= worst, f"{what}: worst frame relative error {ratio.max():.3e} > {worst:.1e} (frame {ratio.argmax()})"
    out = {"rel_rms": total, "worst_frame": float(ratio.max())}
    if ref_s is not None:
        d = np.abs(np.asarray(ref_s, dtype=np.int32).reshape(-1, 160) - np.asarray(got_s, dtype=np.int32).reshape(-1, 160))
        frac = float(np.mean(d <= 1))
        clip = clipped_frames(ref_f)
        below = int(d[~clip].max()) if (~clip).any() else 0
        inside = int(d[clip].max()) if clip.any() else 0
        assert below <= INT16_MAX_LSB, f"{what}: int16 differs by {below} LSB in a frame below the clip"
        assert inside <= INT16_MAX_LSB_CLIPPED, f"{what}: int16 differs by {inside} LSB in a clipped frame"
        assert frac >= 0.999, f"{what}: only {frac:.5f} of int16 samples within 1 LSB"
        assert float(np.mean(clip)) <= MAX_CLIPPED_SHARE, f"{what}: {np.mean(clip):.2f} of the frames reach the soft clip -- too many for the clipped-frame bound to govern"
        out["int16_exact"] = float(np.mean(d == 0))
        out["int16_max"] = int(d.max())
        out["int16_max_below_clip"] = below
        out["clipped_frames"] = float(np.mean(clip))
    return out


def check_results(ref, got, what="results"):
    for name in ("c0_errors", "protected_errors", "c4_errors", "total_errors", "flags"):
        bad = np.nonzero(ref[name] != got[name])[0]
        assert bad.size == 0, f"{what}: {name} differs at frames {bad[:8]} ref={ref[name][bad[:4]]} got={got[name][bad[:4]]}"


def check_state(ref, got, rel=STATE_REL_RMS, what="state"):
    """ref/got: arrays of PARMS_DTYPE with the same shape."""
    ref = np.asarray(ref)
    got = np.asarray(got)
    for name in INT_FIELDS:
        if not np.array_equal(ref[name], got[name]):
            idx = np.argwhere(ref[name] != got[name])[0]
            raise AssertionError(f"{what}: integer field {name} differs first at {tuple(idx)}: ref={ref[name][tuple(idx)]} got={got[name][tuple(idx)]}")
    for name in EXACT_FLOAT_FIELDS:
        a, b = ref[name].view(np.uint32), got[name].view(np.uint32)
        assert np.array_equal(a, b), f"{what}: {name} (integer-valued noise state) differs"
    out = {}
    for name in FLOAT_FIELDS:
        r = rel_rms(ref[name], got[name])
        assert r <= rel, f"{what}: float field {name} relative RMS {r:.3e} > {rel:.1e}"
        out[name] = r
    return out
