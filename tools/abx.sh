#!/bin/bash
# abx.sh <rounds> <workload,...> <variant names...> -- GPU box: interleaved A/B of variant libraries (mbelib-neo_amd/variants/
# libmbx_hip_<name>.so; "product" = the library in the tree) on one box, `rounds` alternations, median kernel time per variant.
# Boxes differ by +-4 %: only numbers from one call compare.  (Development aid.)
cd "$(dirname "$0")/.."
R=$1; WL=$2; shift 2
python - "$R" "$WL" "$@" <<'PY'
import json, os, statistics, subprocess, sys
rounds, wls, names = int(sys.argv[1]), sys.argv[2].split(","), sys.argv[3:]
res = {}
for r in range(rounds):
    for w in wls:
        for n in names:
            env = dict(os.environ)
            if n != "product":
                env["MBX_HIP_LIBRARY"] = os.path.join(os.getcwd(), "mbelib-neo_amd", "variants", f"libmbx_hip_{n}.so")
                env["MBX_HIP_LIBRARY_ALLOW_OLDER"] = "1"
            out = subprocess.run([sys.executable, "bench.py", "--workload", w, "--steps", "10", "--no-cpu-baseline", "--no-extras"],
                                 env=env, capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                res.setdefault((w, n), []).append(d["roofline"]["kernel_ms"])
            except Exception as e:
                print(w, n, "FAILED", out.stderr[-400:])
                sys.exit(1)
for w in wls:
    base = statistics.median(res[(w, names[0])])
    for n in names:
        v = res[(w, n)]
        print(f"{w:12s} {n:10s} median {statistics.median(v):.4f} ms  min {min(v):.4f}  ({statistics.median(v) / base - 1:+.2%} vs {names[0]})  {[round(x, 4) for x in v]}")
PY
