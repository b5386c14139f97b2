#!/bin/bash
# tools/ab.sh "<bench args>" "ENV_A" "ENV_B" [reps]: alternating bench runs, value and ms per step of each
args="$1"; A="$2"; B="$3"; R=${4:-3}
one() { env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-streaming --steps 100 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%-40s' % '$1', d['value'], d['ms_per_step'], 'front %.3f hyst %.3f final %.3f points %.3f lines %.3f' % (s['front'],s['hyst'],s['final'],s['points'],s['lines']))"; }
for i in $(seq $R); do one "$A"; one "$B"; done
