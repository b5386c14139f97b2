#!/bin/bash
# The configurations of the recipe table (throughput by recipe, size, batch and slots), one JSON line each (tools/recipe_probe.py).
out=${1:-gpurun_out/recipes.jsonl}
mkdir -p $(dirname $out)
: > $out
run() { env "$@" 2>/dev/null | tail -1 >> $out; }
run PROBE_SLOTS=4 python tools/recipe_probe.py simple 960 540 256 80
run PROBE_SLOTS=6 python tools/recipe_probe.py curve 1280 720 256 80
run PROBE_SLOTS=8 python tools/recipe_probe.py curve 1280 720 32 300
run PROBE_SLOTS=16 python tools/recipe_probe.py curve 1280 720 32 300
run PROBE_SLOTS=24 python tools/recipe_probe.py curve 1280 720 32 300
run PROBE_SLOTS=6 python tools/recipe_probe.py north_star 1920 1080 256 80
run PROBE_SLOTS=6 python tools/recipe_probe.py hsv_gated 1920 1080 256 80
run PROBE_SLOTS=6 python tools/recipe_probe.py north_star 3840 2160 64 40
run PROBE_SLOTS=9 python tools/recipe_probe.py north_star 3840 2160 64 40
run PROBE_SLOTS=12 python tools/recipe_probe.py north_star 3840 2160 64 40
run PROBE_SLOTS=4 python tools/recipe_probe.py north_star 3840 2160 256 16
run PROBE_SLOTS=3 python tools/recipe_probe.py f1 640 360 256 100
run PROBE_SLOTS=3 python tools/recipe_probe.py f1 640 360 256 60 1920 1080
run PROBE_SLOTS=3 python tools/recipe_probe.py f1 640 360 256 60 1280 720
cat $out
