#!/bin/bash
# abenv.sh <rounds> <workload,...> <name:VAR=val[,VAR=val]>... -- GPU box: interleaved A/B of ONE library under different environments
# (e.g. "staged:MBX_FUSE_ONE=0" "fused:MBX_FUSE_ONE=1"; "name:" alone = the plain environment; a variant may also name a library:
# MBX_HIP_LIBRARY=...), `rounds` alternations, median step and kernel time per variant.  Boxes differ by +-4 %: only numbers from
# one call compare.  (Development aid.)
cd "$(dirname "$0")/.."
R=$1; WL=$2; shift 2
python - "$R" "$WL" "$@" <<'PY'
import json, os, statistics, subprocess, sys
rounds, wls, variants = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
res = {}
for r in range(rounds):
    for w in wls:
        for v in variants:
            name, _, assigns = v.partition(":")
            env = dict(os.environ)
            for a in filter(None, assigns.split(",")):
                k, _, val = a.partition("=")
                env[k] = val
            if "MBX_HIP_LIBRARY" in env:
                env["MBX_HIP_LIBRARY_ALLOW_OLDER"] = "1"
            out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--steps", "10", "--no-cpu-baseline", "--no-extras"],
                                 env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res.setdefault((w, name), []).append((d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel"]))
            except Exception:
                print(w, name, "FAILED", out.stderr[-600:])
                sys.exit(1)
first = variants[0].partition(":")[0]
for w in wls:
    base = statistics.median(x[0] for x in res[(w, first)])
    for v in variants:
        n = v.partition(":")[0]
        x = res[(w, n)]
        st = statistics.median(a[0] for a in x)
        print(f"{w:14s} {n:10s} step median {st:.4f} ms ({st / base - 1:+.2%} vs {first})  kernel median {statistics.median(a[1] for a in x):.4f} ms"
              f"  [{x[0][2]}]  steps {[round(a[0], 4) for a in x]}")
PY
