#!/bin/bash
# One measurement cycle after a library change (the source hash in bdg_version() changes with every edit, and bench.py prints
# roofline.traffic / int_issue only when profiles/traffic.json was taken from the running build):
#
#   here (build container):   tools/round_profile.sh gpu <tag>        # two gpurun calls: tests + profile passes, then bench lines
#                             tools/round_profile.sh collect <tag> [<old tag>]   # condense into profiles/, drop the old tag's files
#
# <tag> e.g. r03_a.  The docs (README.md, BASELINE.md, profiles/README.md) still have to follow by hand.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
mode=$1; tag=$2
GPURUN=/usr/local/graft/bin/gpurun
case $mode in
gpu)
    rm -rf gpurun_out/${tag} gpurun_out/${tag}_c3 gpurun_out/${tag}_c5 gpurun_out/${tag}_ops
    $GPURUN --timeout 1200 -- "timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.txt 2>&1 && bash tools/pmc_profile.sh ${tag} --no-graph && bash tools/pmc_profile.sh ${tag}_c5 --config 5 && bash tools/pmc_profile.sh ${tag}_c3 --config 3 && bash tools/ops_profile.sh ${tag}_ops; tail -1 gpurun_out/${tag}_tests.txt"
    python3 tools/summarize_profile.py gpurun_out/${tag} ${tag} gpurun_out/${tag}_c5 gpurun_out/${tag}_c3   # writes profiles/traffic.json (+ valu, lds) for this build BEFORE the bench lines
    $GPURUN --timeout 1200 -- "python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && python bench.py --config 3 > gpurun_out/${tag}_bench_c3.json 2>> gpurun_out/${tag}_bench.err && python bench.py --config 5 > gpurun_out/${tag}_bench_c5.json 2>> gpurun_out/${tag}_bench.err && python bench.py --no-overlap --no-graph > gpurun_out/${tag}_bench_ov.json 2>> gpurun_out/${tag}_bench.err && python tools/stress_parity.py --cases 2 > gpurun_out/${tag}_stress.txt 2>&1; tail -n 1 gpurun_out/${tag}_stress.txt"
    ;;
collect)
    old=${3:-}
    if [ -n "$old" ]; then git rm -q --cached profiles/${old}_* 2>/dev/null || true; rm -f profiles/${old}_*; fi
    python3 tools/summarize_profile.py gpurun_out/${tag} ${tag} gpurun_out/${tag}_c5 gpurun_out/${tag}_c3
    for t in c5 c3 ops; do cp "$(find gpurun_out/${tag}_$t/stats -name '*_kernel_stats.csv' | head -1)" profiles/${tag}_${t}_kernel_stats.csv; done
    cp "$(find gpurun_out/${tag}/stats -name '*_kernel_stats.csv' | head -1)" profiles/${tag}_kernel_stats.csv
    cp gpurun_out/${tag}_ops/ops.jsonl profiles/${tag}_ops.jsonl
    round=${tag%%_*}
    cp gpurun_out/${tag}_bench.json profiles/${round}_bench.json
    cp gpurun_out/${tag}_bench_c3.json profiles/${round}_bench_config3.json
    cp gpurun_out/${tag}_bench_c5.json profiles/${round}_bench_config5.json
    cp gpurun_out/${tag}_bench_ov.json profiles/${round}_bench_no_overlap.json
    python3 - "$tag" "$round" <<'PY'
import json, sys
tag, rnd = sys.argv[1], sys.argv[2]
d = json.load(open("profiles/%s_summary.json" % tag))
for k in ("k_scan_reads", "k_sw_clusters", "k_strict_filter", "k_finalize_reads", "k_nearest_pairs", "k_nearest_delins"):
    v = d[k]
    print(k, "launches", v["calls"], round(v["avg_ns"] / 1e3, 1), "us", round(v.get("SQ_INSTS_VALU", 0) / 1e6, 1), "M vector instructions")
for f in ("bench", "bench_config3", "bench_config5", "bench_no_overlap"):
    x = json.loads(open("profiles/%s_%s.json" % (rnd, f)).read().strip().splitlines()[-1])
    print(f, round(x["value"] / 1e6, 1), x["unit"], round(x["ms_per_step"], 4), "ms", x["parity_sample"], x["roofline"]["kernel"],
          round(x["roofline"]["kernel_ms"], 4), "frac", round(x["roofline"]["frac"], 4), "traffic", x["roofline"].get("traffic"))
PY
    ;;
*) echo "usage: $0 gpu|collect <tag> [<old tag>]"; exit 2;;
esac
