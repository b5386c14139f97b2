#!/bin/bash
# bench variants for the HoughLinesP co-residency experiment: tools/ppht_try.sh "<bench args>" VAR=val ...
args="$1"; shift
echo "== $args $*"; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-streaming --steps 60 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d['stage_ms'].items()}, 'excl lines', d['roofline_exclusive']['stage_ms']['lines'], 'shared', d['streams_sharing_a_queue'])"
