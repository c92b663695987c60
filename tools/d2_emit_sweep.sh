#!/bin/bash
# builder tool: the deletion-variant join at different numbers of resident blocks per compute unit (index passes / pair kernel)
for b in 2 4 6 8; do echo "pairs blocks $b"; BADGER_AMD_D2_PAIRS_BLOCKS=$b GRAPH_SIZES_QJOIN_MAX=0 python tools/graph_sizes.py 500000 4000000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['rows'], d['algo5_ms'], d['algo5_kernels_ms'])"; done
