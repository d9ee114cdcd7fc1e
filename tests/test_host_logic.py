"""CPU suite for the host side: C-ABI exports, packers, frame generator, state constructors,
stream sharding and the table broadcast (gloo, world_size 2).  No GPU compute here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    import mbelib_neo_amd as m
    from mbelib_neo_amd import _native

    header = open(os.path.join(ROOT, "include", "mbx.h")).read()
    declared = set(re.findall(r"\b(mbx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found in include/mbx.h"
    assert os.path.exists(m.library_path()), "libmbx_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    try:
        handle = C.CDLL(m.library_path())
    except OSError as e:
        pytest.skip(f"HIP runtime not loadable here: {e}")
    for name in sorted(declared):
        assert hasattr(handle, name), f"libmbx_hip.so does not export {name}"
    assert declared == set(_native.EXPORTED_SYMBOLS)
    # no fault-injection or debug hook in the shipping ABI: those live in the -DMBX_TESTING / -DMBX_ABLATE builds only
    import subprocess

    exported = subprocess.run(["nm", "-D", "--defined-only", m.library_path()], capture_output=True, text=True).stdout
    assert "mbx_process_batch" in exported
    assert "mbx_testing_" not in exported and "mbx_debug_set_" not in exported


def test_launchers_fail_loudly_without_init_or_device():
    from mbelib_neo_amd import _native

    try:
        L = _native.lib()
    except _native.NativeLibraryError as e:
        pytest.skip(str(e))
    # no device / no mbx_init() for the current device: every launcher refuses (no silent CPU path)
    import torch

    refusal = -101 if torch.cuda.is_available() else -100   # MBX_ENOTINIT with a device, MBX_ENODEVICE without
    assert L.mbx_fec_imbe7200x4400(None, 1, None, None) == refusal
    assert L.mbx_process_records(0, 1, 1, None, None, None, None, None, None, None) == refusal
    assert L.mbx_process_records_ws(0, 1, 1, None, None, None, None, None, None, None, 0, None) == refusal
    assert L.mbx_reserve(16) == refusal and L.mbx_device_ready(0) == 0
    assert L.mbx_last_error()   # a message is left for the calling thread
    blob = open(os.path.join(ROOT, "mbelib-neo_amd", "data", "mbx_tables.bin"), "rb").read()

    if not torch.cuda.is_available():
        assert L.mbx_init(0, blob, len(blob)) == -100  # MBX_ENODEVICE
    bad = bytearray(blob)
    bad[5000] ^= 1
    assert L.mbx_init(0, bytes(bad), len(bad)) == -102  # checksum


def test_packers_match_oracle_and_validate(oracle):
    from mbelib_neo_amd import _native

    try:
        L = _native.lib()
    except _native.NativeLibraryError as e:
        pytest.skip(str(e))
    rng = np.random.default_rng(1)
    for codec, ncell, fn in ((0, 184, L.mbx_pack_imbe7200x4400), (1, 96, L.mbx_pack_ambe3600x2450)):
        cells = rng.integers(0, 2, size=(64, ncell), dtype=np.int8)
        _, ref = oracle.pack(codec, cells)
        got = np.zeros_like(ref)
        assert fn(cells.ctypes.data, 64, got.ctypes.data) == 0
        assert np.array_equal(ref, got)
        cells[7, ncell - 1] = -1
        untouched = np.full_like(ref, 0x55)
        assert fn(cells.ctypes.data, 64, untouched.ctypes.data) == -2 and (untouched == 0x55).all()
        assert fn(None, 1, got.ctypes.data) == -1


def test_state_constructors_match_oracle(oracle):
    from mbelib_neo_amd.layout import init_state, rng_default, rng_seeded

    assert init_state(3).tobytes() == oracle.init_state(3).tobytes()
    seeds = [0, 1, 1234, 0xC0FFEE, 0xFFFFFFFF]
    assert rng_seeded(seeds).tobytes() == oracle.rng_seeded(seeds).tobytes()
    assert rng_default(2).tobytes() == oracle.rng_default(2).tobytes()


def test_framegen_encoders_roundtrip_through_oracle_fec(oracle):
    from mbelib_neo_amd import framegen

    rng = framegen.rng_for(5)
    bits = rng.integers(0, 2, size=(512, 88), dtype=np.uint8)
    rec = oracle.fec_batch(0, framegen.encode_imbe7200x4400(bits))
    assert np.array_equal(oracle_lib.records_to_bits(rec, 88), bits)
    assert int(oracle_lib.records_to_results(rec)["total_errors"].max()) == 0
    bits = rng.integers(0, 2, size=(512, 49), dtype=np.uint8)
    rec = oracle.fec_batch(1, framegen.encode_ambe3600x2450(bits))
    assert np.array_equal(oracle_lib.records_to_bits(rec, 49), bits)
    assert int(oracle_lib.records_to_results(rec)["total_errors"].max()) == 0
    # up to 3 flips per Golay word / 1 per Hamming word are corrected back to the same bits
    frames = framegen.encode_imbe7200x4400(bits := rng.integers(0, 2, size=(256, 88), dtype=np.uint8))
    noisy = frames.copy()
    noisy[:, 0] ^= 0x41  # two flips in row 0
    noisy[:, 12] ^= 0x08  # one flip in a Hamming row
    rec = oracle.fec_batch(0, noisy)
    assert np.array_equal(oracle_lib.records_to_bits(rec, 88), bits)
    assert (oracle_lib.records_to_results(rec)["total_errors"] >= 1).all()


def test_voiced_workload_is_all_voiced(oracle):
    from mbelib_neo_amd import framegen

    S = 256
    rng = framegen.rng_for(2)
    frames = np.concatenate([framegen.imbe_clean_voiced_frames(S, rng)[:, None, :]] * 2, axis=1).reshape(S * 2, 18)
    out = oracle.process_batch(0, S, 2, frames, oracle.init_state(S), oracle.rng_seeded(range(S)))
    st = out["state"]
    for s in range(S):
        L = int(st[s, 1]["L"])
        assert (st[s, 1]["Vl"][1 : L + 1] == 1).all()  # pre-enhancement snapshot: every band voiced
    assert int(out["results"]["total_errors"].max()) == 0 and int(out["results"]["flags"].max()) == 0x06


def test_shard_range_partitions_streams():
    from mbelib_neo_amd.parallel import shard_range

    for total, world in ((65536, 8), (10, 3), (7, 8), (0, 4)):
        got = [shard_range(total, world, r) for r in range(world)]
        assert sum(c for _, c in got) == total
        nxt = 0
        for first, count in got:
            assert first == nxt
            nxt += count


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import mbelib_neo_amd as m
from mbelib_neo_amd.parallel import broadcast_tables, shard_range, blob_checksum
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
blob = broadcast_tables(m.load_tables_blob() if rank == 0 else None)
assert blob == m.load_tables_blob()
cs = torch.tensor([blob_checksum(blob)], dtype=torch.int64)
got = [torch.zeros_like(cs) for _ in range(world)]
dist.all_gather(got, cs)
assert all(int(g) == int(cs) for g in got)
# sharded decode of 8 streams with the ORACLE standing in for the device (test only): the union
# of the shards must equal the unsharded result, i.e. sharding needs no exchange step
sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle_lib
from mbelib_neo_amd import framegen
o = oracle_lib.load()
S, T = 8, 3
frames = framegen.random_frames(0, S * T, framegen.rng_for(11)).reshape(S, T, 18)
first, count = shard_range(S, world, rank)
mine = o.process_batch(0, count, T, frames[first:first + count].reshape(-1, 18), o.init_state(count),
                       o.rng_seeded(np.arange(first, first + count) + 1234))
h = torch.tensor([o.fnv(mine["pcm16"])], dtype=torch.int64)
hs = [torch.zeros_like(h) for _ in range(world)]
dist.all_gather(hs, h)
if rank == 0:
    full = o.process_batch(0, S, T, frames.reshape(-1, 18), o.init_state(S), o.rng_seeded(np.arange(S) + 1234))
    parts = [full["pcm16"].reshape(S, T, 160)[f:f + c] for f, c in (shard_range(S, world, r) for r in range(world))]
    assert [int(x) for x in hs] == [o.fnv(np.ascontiguousarray(p)) for p in parts]
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_table_broadcast_and_sharding_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_shim_library_exports_every_declared_symbol():
    import shim_lib

    assert os.path.exists(shim_lib.PATH), "libmbe_neo_amd.so not built"
    names = shim_lib.declared_symbols()
    assert len(names) >= 30
    try:
        handle = C.CDLL(shim_lib.PATH)
    except OSError as e:
        pytest.skip(f"HIP runtime not loadable here: {e}")
    for name in names:
        assert hasattr(handle, name), f"libmbe_neo_amd.so does not export {name}"


def test_shim_library_exports_the_whole_reference_api():
    """tests/golden/mbelib_api_symbols.txt = the 87 function names of the reference's public header
    (include/mbelib-neo/mbelib.h, listed by oracle/tools/list_api_symbols.sh): a host linked against libmbe-neo.so.2
    must find every one of them in libmbe_neo_amd.so."""
    import shim_lib

    want = open(os.path.join(ROOT, "tests", "golden", "mbelib_api_symbols.txt")).read().split()
    assert len(want) == 87
    out = subprocess.check_output(["nm", "-D", "--defined-only", shim_lib.PATH]).decode()
    have = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in want if n not in have]
    assert not missing, f"libmbe_neo_amd.so does not export: {missing}"
    assert set(want) <= set(shim_lib.declared_symbols())   # and include/mbe_neo_amd.h declares them


def test_bench_self_launch_starts_ranks_and_relays_rank0(tmp_path):
    """bench.py --gpus N without a launcher environment starts the N ranks itself (before any GPU call): every child gets
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0's output is relayed, the worst exit code is returned."""
    sys.path.insert(0, ROOT)
    import bench

    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys\n"
        "r = int(os.environ['RANK'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert int(os.environ['MASTER_PORT']) > 0 and 'torch' not in sys.modules\n"
        "print('{\"rank\": %d, \"args\": \"%s\"}' % (r, ' '.join(sys.argv[1:])))\n"
        "sys.exit(3 if (r == 2 and '--fail' in sys.argv) else 0)\n")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(sys.argv[1:], 3, script=%r))" % (ROOT, str(script)))
    ok = subprocess.run([sys.executable, "-c", code, "--gpus", "3", "--steps", "7"], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and ok.stdout.strip() == '{"rank": 0, "args": "--gpus 3 --steps 7"}', ok.stdout + ok.stderr
    bad = subprocess.run([sys.executable, "-c", code, "--fail"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 3


def test_program_written_against_the_reference_header_links(tmp_path):
    """Drop-in at link level: a C program that includes the REFERENCE's own public header (where the reference tree is
    present, i.e. in the authoring container; skipped elsewhere) and calls one function of every family links against
    libmbe_neo_amd.so with no undefined symbol and no prototype conflict."""
    import shim_lib

    ref_inc = "/root/reference/include"
    gen_inc = os.path.join(ROOT, "oracle", "_ref", "include")   # the generated one-line version.h (oracle/Makefile)
    if not os.path.exists(os.path.join(ref_inc, "mbelib-neo", "mbelib.h")) or not os.path.exists(os.path.join(gen_inc, "mbelib-neo", "version.h")):
        pytest.skip("reference tree not present here")
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <mbelib-neo/mbelib.h>
int main(void) {
    static mbe_parms cur, prev, enh;
    static char imbe_fr[8][23], imbe7100_fr[7][24], ambe_fr[4][24], imbe_d[88], ambe_d[49];
    static mbe_soft_bit soft_i[8][23], soft_a[4][24];
    static short pcm[160];
    static float pcmf[160];
    mbe_process_result res;
    long block = 0;
    mbe_initMbeParms(&cur, &prev, &enh);
    mbe_initProcessResult(&res);
    mbe_setThreadRngSeed(1u);
    int n = 0;
    n += mbe_eccImbe7200x4400C0(imbe_fr) + mbe_demodulateImbe7200x4400Data(imbe_fr) + mbe_eccImbe7200x4400Data(imbe_fr, imbe_d);
    n += mbe_decodeImbe4400Parms(imbe_d, &cur, &prev);
    n += mbe_processImbe7200x4400Frame(pcm, &res, (const char(*)[23])imbe_fr, imbe_d, &cur, &prev, &enh);
    n += mbe_processImbe7200x4400SoftFramef(pcmf, &res, (const mbe_soft_bit(*)[23])soft_i, imbe_d, &cur, &prev, &enh);
    n += mbe_processImbe7100x4400Framef(pcmf, &res, (const char(*)[24])imbe7100_fr, imbe_d, &cur, &prev, &enh);
    n += mbe_convertImbe7100to7200(imbe_d);
    n += mbe_processAmbe3600x2450Frame(pcm, &res, (const char(*)[24])ambe_fr, ambe_d, &cur, &prev, &enh);
    n += mbe_processAmbe3600x2450SoftFrame(pcm, &res, (const mbe_soft_bit(*)[24])soft_a, ambe_d, &cur, &prev, &enh);
    n += mbe_processAmbe3600x2400Framef(pcmf, &res, (const char(*)[24])ambe_fr, ambe_d, &cur, &prev, &enh);
    n += mbe_decodeAmbe2450Parms(ambe_d, &cur, &prev) + mbe_decodeAmbe2400Parms(ambe_d, &cur, &prev);
    n += mbe_eccAmbe3600x2450C0(ambe_fr) + mbe_demodulateAmbe3600x2400Data(ambe_fr);
    n += mbe_checkGolayBlock(&block) + mbe_golay2312(imbe_fr[0], imbe_fr[1]) + mbe_hamming1511(imbe_fr[4], imbe_fr[5]);
    mbe_synthesizeSpeechf(pcmf, &cur, &prev);
    mbe_spectralAmpEnhance(&cur);
    mbe_applyAdaptiveSmoothing(&cur, &prev);
    mbe_floattoshort(pcmf, pcm);
    mbe_synthesizeComfortNoise(pcm);
    mbe_synthesizeTonef(pcmf, ambe_d, &cur);
    mbe_dumpImbe7200x4400Frame((const char(*)[23])imbe_fr);
    return n + (mbe_versionString() != 0);
}
''')
    exe = tmp_path / "host"
    libdir = os.path.dirname(shim_lib.PATH)
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", ref_inc, "-I", gen_inc, str(src), "-o", str(exe), "-L", libdir, "-lmbe_neo_amd",
                        "-Wl,--no-undefined", "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_fft_swizzle_of_the_kernel_is_conflict_free_in_the_bank_model():
    """The LDS-resident stream kernels address the unvoiced transform through an XOR swizzle (mbx_stream.hip, fsw).  Under
    the gfx950 banking rules of MI355X_MICROARCH.md (ds_read_b64: 2 x 32 lanes on 64 dword banks; ds_write_b64: 4 x 16 lanes
    on 32) every access pattern of the transform pair is conflict-free with it (200 lane-group cycles = the minimum), the
    plain indices are not (504), and the map is a bijection of 0..255 that is linear over GF(2) -- which is what lets the
    kernel form fsw(base + r q) as fsw(base) ^ fsw(r q)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fft_swizzle", os.path.join(ROOT, "tools", "fft_swizzle.py"))
    fs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fs)
    assert fs.cost(lambda e: e) == (504, 200)
    assert fs.cost(fs.kernel_map) == (200, 200)
    assert sorted(fs.kernel_map(e) for e in range(256)) == list(range(256))
    for a in range(256):
        for b in (1, 2, 3, 4, 8, 12, 16, 32, 48, 64, 128, 192):
            assert fs.kernel_map(a ^ b) == fs.kernel_map(a) ^ fs.kernel_map(b)
    src = open(os.path.join(ROOT, "mbelib-neo_amd", "csrc", "mbx_stream.hip")).read()
    assert "(e ^ ((e >> 2) & 3) ^ (((e >> 4) & 7) << 2))" in src   # the kernel's map is the one checked here


def test_exact_shortcuts_of_the_stream_kernel_are_exact():
    """Two float shortcuts of mbx_stream.hip claim to be BIT-identical to what the reference computes (round 4; each replaces a
    12-36-instruction IEEE expansion executed once per frame).  Their arithmetic is restated here in numpy -- float32 values,
    the FMA's single rounding emulated in float64, which is exact for these operand widths -- and checked against the plain
    expressions on millions of values:
      wrap_two_pi(x)            = fmodf(x, 2 pi)              for 0 <= x < 4e6   (ref src/core/mbelib.c:901-912)
      div_by_uniform(a, L, 1/L) = a / L  correctly rounded    for L = 1..56       (ref src/core/mbelib.c:925-940, the phase offset)"""
    rng = np.random.default_rng(20240404)
    y = np.float32(2.0) * np.float32(np.pi)
    c = np.float32(np.float32(0.15915494) * np.float32(1.000001))

    def wrap(x):
        n = np.trunc(x * c).astype(np.float32)
        r = x.astype(np.float64) - n.astype(np.float64) * np.float64(y)     # the FMA: exact product, one rounding ...
        r32 = r.astype(np.float32)
        assert np.all(r32.astype(np.float64) == r)                          # ... which never rounds: the value is a float
        r2 = np.where(r32 < 0, r32.astype(np.float64) + np.float64(y), r32.astype(np.float64))
        out = r2.astype(np.float32)
        assert np.all(out.astype(np.float64) == r2)
        return out

    for scale in (7.0, 100.0, 5000.0, 1e5, 3.9e6):
        x = (rng.random(1_000_000) * scale).astype(np.float32)
        assert np.array_equal(np.fmod(x, y).view(np.uint32), wrap(x).view(np.uint32)), scale
    k = np.arange(0, 600000, dtype=np.float64)
    edge = (k * np.float64(y)).astype(np.float32)
    for x in (edge, np.nextafter(edge, np.float32(np.inf)), np.nextafter(edge, np.float32(0))):
        x = np.abs(x)
        assert np.array_equal(np.fmod(x, y).view(np.uint32), wrap(x).view(np.uint32))
    for L in range(1, 57):
        b = np.float32(L)
        rcp = np.float32(1) / b
        a = ((rng.random(100_000) * 2 - 1) * np.float32(np.pi) * rng.integers(1, 57, 100_000)).astype(np.float32)
        q = (a * rcp).astype(np.float32)
        r = a.astype(np.float64) - q.astype(np.float64) * np.float64(b)
        got = (r * np.float64(rcp) + q.astype(np.float64)).astype(np.float32)
        ref = (a.astype(np.float64) / np.float64(b)).astype(np.float32)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), L
    src = open(os.path.join(ROOT, "mbelib-neo_amd", "csrc", "mbx_stream.hip")).read()
    assert "truncf(x * (0.15915494f * 1.000001f))" in src and "return fmaf(r, rcp_b, q);" in src   # the kernel's expressions are these


def test_single_frame_fec_closed_form_sequence_and_header_gather():
    """Two restatements inside the kernels that the GPU tests exercise only through whole frames, checked here on their own:
    (1) the single-frame FEC's demodulation sequence in closed form (mbx_fec_frame.h, PrLane / PrWave: lane j holds (A, C) of
        steps j + 1 and j + 65 by seven doublings; x_k = A_k x_0 + C_k mod 2^16; mask_for() cuts a row's mask out of the two
        64-bit ballots, first bit to the top) against the serial recurrence x_k = 173 x_{k-1} + 13849 mod 2^16
        (ref src/imbe/imbe7200x4400.c:469-515, src/ambe/ambe_common.c:127-157) for all 4,096 seeds and the three codecs' row widths;
    (2) the lane -> dword map of the gathered header load / store (mbx_stream.hip, load_header / store_parms<kGather>) against the
        byte offsets of the fourteen scalar fields of mbe_parms (include/mbx_types.h layout via mbelib_neo_amd.layout)."""
    def serial(seed, n=128):
        x = (16 * seed) & 0xFFFF
        out = []
        for _ in range(n):
            x = (173 * x + 13849) & 0xFFFF
            out.append(x >> 15)
        return out

    def lane_consts(lane):
        P, Q, a, c = 173, 13849, 1, 0
        k = lane + 1
        for b in range(7):
            if k & (1 << b):
                a, c = (a * P) & 0xFFFF, (c * P + Q) & 0xFFFF
            Q, P = (Q * (P + 1)) & 0xFFFF, (P * P) & 0xFFFF
        P64, Q64 = 173, 13849
        for b in range(6):
            Q64, P64 = (Q64 * (P64 + 1)) & 0xFFFF, (P64 * P64) & 0xFFFF
        return a, c, (a * P64) & 0xFFFF, (c * P64 + Q64) & 0xFFFF

    lanes = [lane_consts(l) for l in range(64)]

    def brev32(v):
        return int(f"{v & 0xFFFFFFFF:032b}"[::-1], 2)

    for seed in range(4096):
        x0 = (16 * seed) & 0xFFFF
        lo = sum(((((a1 * x0 + c1) >> 15) & 1) << j) for j, (a1, c1, _, _) in enumerate(lanes))
        hi = sum(((((a2 * x0 + c2) >> 15) & 1) << j) for j, (_, _, a2, c2) in enumerate(lanes))
        bits = serial(seed)
        assert [(lo >> j) & 1 for j in range(64)] + [(hi >> j) & 1 for j in range(64)] == bits, seed
        if seed % 37 == 0:
            for widths in ((23, 23, 23, 15, 15, 15), (24, 23, 23, 15, 15), (23,)):
                pos = 0
                for w in widths:
                    cut = ((lo >> pos) | ((hi << (64 - pos)) if pos > 0 else 0)) if pos < 64 else (hi >> (pos - 64))
                    got = brev32(cut & 0xFFFFFFFF) >> (32 - w)
                    want = 0
                    for j in range(w):
                        want = (want << 1) | bits[pos + j]
                    assert got == want, (seed, widths, pos)
                    pos += w
    src = open(os.path.join(ROOT, "mbelib-neo_amd", "csrc", "mbx_fec_frame.h")).read()
    assert "a = (a * P) & 0xffffu;" in src and "c = (c * P + Q) & 0xffffu;" in src and "return __brev((uint32_t)cut) >> (32 - width);" in src

    from mbelib_neo_amd.layout import PARMS_DTYPE
    names = ["w0", "L", "K", "gamma", "tonePhase", "swn", "localEnergy", "amplitudeThreshold", "errorRate", "errorCountTotal",
             "errorCount4", "repeatCount", "mutingThreshold", "noiseSeed"]
    fields = {n.lower(): PARMS_DTYPE.fields[n][1] for n in PARMS_DTYPE.names}

    def offset_of(name):   # the layout module may spell a field differently from the kernel's Parms: match loosely
        key = name.lower()
        cands = [k for k in fields if k == key] or [k for k in fields if k.replace("_", "") == key.replace("_", "")]
        assert cands, (name, sorted(fields))
        return fields[cands[0]]

    for j, name in enumerate(names):
        idx = j if j < 3 else ((288 - 3) + j if j < 13 else 554)   # load_header's map (O_GAMMA = 288, O_NOISESEED = 554)
        assert offset_of(name) == 4 * idx, (name, offset_of(name), idx)
    ksrc = open(os.path.join(ROOT, "mbelib-neo_amd", "csrc", "mbx_stream.hip")).read()
    assert "(j < 3) ? j : ((j < 13) ? (O_GAMMA - H_GAMMA) + j : O_NOISESEED)" in ksrc and "O_GAMMA = 288" in ksrc and "O_NOISESEED = 554" in ksrc


def test_bench_contract_line_is_compact_and_complete():
    """bench.contract_line(): the ONE stdout line the driver parses, built from a canned full measurement (tests/golden/bench_detail_canned.json:
    round 5's 20 KB line, the one the driver could not parse): every contract key, roofline / parity / cpu_baseline, five numbers per other
    config, at most 4,096 bytes -- also when the free-text fields grow."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    detail = json.load(open(os.path.join(root, "tests", "golden", "bench_detail_canned.json")))
    assert len(json.dumps(detail)) > 15000
    line = bench.contract_line(detail)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["value"] == float(f"{detail['value']:.7g}") and line["config"]["workload"].startswith("BASELINE configs[1]") and "model" not in line["config"]
    assert {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms"} <= set(line["roofline"])
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-12
    assert {"rel_rms", "worst_frame", "int16_within_1", "int16_max", "streams_checked", "results_exact"} <= set(line["parity"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert all(set(v) == {"value", "ms_per_step", "kernel", "kernel_ms", "frac"} for v in line["other_configs"].values()) and len(line["other_configs"]) == 5
    for k in ("issue", "kernel_ms_stats", "copy_floor"):
        assert k not in line["roofline"]
    for k in ("cpu_baselines", "host_path", "valu", "infinity_cache_assisted", "convert", "distributed"):
        assert k not in line
    # growth: long free text is clipped, and in the last resort other_configs goes before any contract key does
    fat = json.loads(json.dumps(detail))
    fat["config"]["workload"] *= 20
    fat["cpu_baseline"]["sample"] *= 20
    fat["parity"]["FAILED"] = "x" * 5000
    fat["other_configs"].update({f"extra{i}": fat["other_configs"]["ambe_fec"] for i in range(40)})
    thin = bench.contract_line(fat)
    assert len(json.dumps(thin)) <= 4096 and "other_configs" not in thin and thin["parity"]["FAILED"].startswith("xxx") and "roofline" in thin and "cpu_baseline" in thin


def test_wire_permutation_from_a_callers_deinterleave_tables():
    """a host folds its own burst -> cell tables into one permutation that writes the packed wire frame
    directly (include/mbx.h, mbx_wire_bit_of_cell / mbx_wire_permutation).  For random schedules (a random bijection from received-bit
    order onto the codec's cells -- the reference holds no air-interface table to test against) and random bits: scattering the bits
    through the permutation gives the bytes mbx_pack_* produce from the cell array the same schedule fills; schedules that name a cell
    twice, leave the frame or have the wrong length are refused.  Host-only: no device needed."""
    from mbelib_neo_amd import _native
    from mbelib_neo_amd.layout import FRAME_BYTES, FRAME_CELLS, ROW_WIDTHS

    try:
        L = _native.lib()
    except _native.NativeLibraryError as e:
        pytest.skip(str(e))
    rng = np.random.default_rng(7)
    packers = {0: L.mbx_pack_imbe7200x4400, 1: L.mbx_pack_ambe3600x2450, 2: L.mbx_pack_imbe7100x4400, 3: L.mbx_pack_ambe3600x2450}
    for codec in (0, 1, 2, 3):
        rows, cols = FRAME_CELLS[codec]
        widths = ROW_WIDTHS[codec]
        cells = [(r, j) for r in range(rows) for j in range(widths[r])]
        n = len(cells)
        assert n == {0: 144, 1: 72, 2: 142, 3: 72}[codec]
        assert L.mbx_wire_bit_of_cell(codec, 0, widths[0] - 1) == 0 and L.mbx_wire_bit_of_cell(codec, 0, 0) == widths[0] - 1
        assert L.mbx_wire_bit_of_cell(codec, rows - 1, 0) == n - 1 and L.mbx_wire_bit_of_cell(codec, 0, widths[0]) == -1
        assert L.mbx_wire_bit_of_cell(codec, rows, 0) == -1 and L.mbx_wire_bit_of_cell(7, 0, 0) == -1
        for _ in range(8):
            order = rng.permutation(n)
            cr = np.ascontiguousarray([cells[k][0] for k in order], dtype=np.int32)
            cc = np.ascontiguousarray([cells[k][1] for k in order], dtype=np.int32)
            wb = np.zeros(n, dtype=np.int32)
            assert L.mbx_wire_permutation(codec, cr.ctypes.data, cc.ctypes.data, n, wb.ctypes.data) == 0
            assert sorted(wb.tolist()) == list(range(n))
            bits = rng.integers(0, 2, size=n)
            frame = np.zeros(FRAME_BYTES[codec], dtype=np.uint8)
            for i in range(n):
                if bits[i]:
                    frame[wb[i] >> 3] |= 0x80 >> (wb[i] & 7)
            arr = np.zeros((rows, cols), dtype=np.int8)   # what the caller's deinterleaver would have filled
            arr[cr, cc] = bits
            want = np.zeros(FRAME_BYTES[codec], dtype=np.uint8)
            assert packers[codec](arr.ctypes.data, 1, want.ctypes.data) == 0
            assert frame.tobytes() == want.tobytes(), codec
        bad = cr.copy()
        bad_c = cc.copy()
        bad[1], bad_c[1] = bad[0], bad_c[0]   # a cell named twice
        assert L.mbx_wire_permutation(codec, bad.ctypes.data, bad_c.ctypes.data, n, wb.ctypes.data) == -1
        assert L.mbx_wire_permutation(codec, cr.ctypes.data, cc.ctypes.data, n - 1, wb.ctypes.data) == -1
        bad_c = cc.copy()
        bad_c[0] = 24   # outside every row
        assert L.mbx_wire_permutation(codec, cr.ctypes.data, bad_c.ctypes.data, n, wb.ctypes.data) == -1
