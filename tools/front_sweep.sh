#!/bin/bash
# unit height / waves per workgroup sweep of the streaming front kernel alone (tools/front_probe.py)
for rows in ${ROWS:-72 108 135 180 216 270 360 540}; do
  for waves in ${WAVES:-1 2 4}; do
    echo -n "rows=$rows waves=$waves: "
    LANECV_FRONT_ROWS=$rows LANECV_FRONT_WAVES=$waves FRONT_PROBE_ITERS=30 timeout -k 10 120 python3 tools/front_probe.py 2>&1 | tr '\n' ' '
    echo
  done
done
