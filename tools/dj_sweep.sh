#!/bin/bash
# deletion-variant joins: bench configs 5 and 3 under a few geometries (gpurun_out/<tag>_*.json), one line each
# usage: tools/dj_sweep.sh <tag> [name:config:ENV=..,ENV=.. ...]
tag=${1:-dj}; shift
run() {  # name, config, env...
    name=$1; cfg=$2; shift 2
    env "$@" timeout -k 10 200 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${tag}_${name}.json").read())
    print("${name}", "%.4f" % d["ms_per_step"], d["kernels_ms_per_step"], d["config"]["edges_rank0"])
except Exception as e:
    print("${name}", "ERR", e, open("gpurun_out/${tag}_${name}.err").read()[-400:])
PY
}
if [ $# -eq 0 ]; then set -- c5:5:X=1 c3:3:X=1; fi
for spec in "$@"; do
    IFS=: read name cfg envs <<< "$spec"
    run $name $cfg ${envs//,/ }
done
