#!/bin/bash
# usage: tools/chunk_sweep.sh <workload> <chunks...> [-- bench args]   samples-per-slab sweep of the gradient kernels (RBNN_CHUNK)
wl=$1; shift
chunks=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do chunks+=($1); shift; done; [ "$1" = "--" ] && shift
for c in "${chunks[@]}"; do echo "chunk $c"; RBNN_CHUNK=$c python bench.py --workload $wl --cpu-seconds 0 --no-other-mode "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v['avg_ms'],3) for k,v in d['roofline']['kernels'].items()})"; done
