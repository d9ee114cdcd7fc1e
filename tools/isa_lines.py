#!/usr/bin/env python3
"""isa_lines.py <kernel-substring> [src=mbx_stream.hip] -- static instruction histogram of one kernel by SOURCE LINE.

Compiles csrc/<src> for gfx950 with the product flags plus -gline-tables-only (device only, no GPU needed), disassembles
with line info and prints, per source line range, how many VALU / transcendental / SALU / LDS / VMEM instructions the
kernel carries there.  Static counts: a loop body counts once (the header of the output says so) -- multiply by the trip
count yourself.  Development aid: where do the instructions of a stage go, before and after a change.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mbelib-neo_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def build(src, extra):
    o = f"/tmp/isa_lines_{os.getpid()}.o"
    elf = o[:-2] + ".elf"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-mllvm", "-disable-machine-licm",
                           "-gline-tables-only", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", o] + extra)
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + o,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    text = subprocess.check_output([LLVM + "/llvm-objdump", "-d", "-l", elf], text=True)
    os.unlink(o)
    os.unlink(elf)
    return text


def classify(op):
    if op.startswith(("v_cos", "v_sin", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    want = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else "mbx_stream.hip"
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    text = build(src, extra)
    cur_kernel, line = None, 0
    hist = collections.defaultdict(collections.Counter)
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            cur_kernel = m.group(1)
            continue
        if cur_kernel is None or want not in cur_kernel or (want + "_lds" in cur_kernel and not want.endswith("_lds")):
            continue
        m = re.match(r"^; \S*?([\w.]+):(\d+)", ln)
        if m:
            line = int(m.group(2)) if m.group(1).startswith(os.path.basename(src).split(".")[0]) else -1
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", ln)
        if m:
            hist[line][classify(m.group(1))] += 1
    tot = collections.Counter()
    for c in hist.values():
        tot.update(c)
    print(f"kernel *{want}*: static totals {dict(tot)}  (loop bodies count once)")
    step = 10
    buckets = collections.defaultdict(collections.Counter)
    for l, c in hist.items():
        buckets[(l // step) * step if l >= 0 else -1].update(c)
    for b in sorted(buckets):
        c = buckets[b]
        print(f"  {('other files' if b < 0 else f'{b:5d}-{b + step - 1:<5d}'):>12}  valu {c['valu']:4d}  trans {c['trans']:3d}  salu {c['salu']:4d}  "
              f"lds {c['lds']:3d}  vmem {c['vmem']:3d}")


if __name__ == "__main__":
    main()
