#!/bin/bash
# profile_all.sh <round> -- GPU box: tools/profile_round.sh for every bench workload that has a line in the default bench run, one after
# the other, summaries into profiles/<round>/ (copy them home from gpurun_out/<round>_*), then the default bench line LAST so that it
# carries the traffic and issue figures of this very build.  About two minutes per workload.
cd "$(dirname "$0")/.."
R=${1:-r06}
mkdir -p gpurun_out
for w in imbe_voiced imbe_voiced_resident ambe_fec ambe_fec_resident imbe_mixed ambe_stream; do
  echo "== $w"; tools/profile_round.sh ${R}_$w $w > gpurun_out/${R}_${w}_profile.log 2>&1; tail -2 gpurun_out/${R}_${w}_profile.log | cut -c1-400
done
python3 bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err && cp bench_detail.json gpurun_out/${R}_bench_default_detail.json
wc -c gpurun_out/${R}_bench_default.json; cat gpurun_out/${R}_bench_default.json
