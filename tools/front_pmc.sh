#!/bin/bash
# SQ counters of the streaming front kernel alone: tools/front_pmc.sh <out-subdir>
out=gpurun_out/${1:-pmc}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc1 -- python3 tools/front_probe.py > $out/pmc1.txt 2>&1
echo rc=$?
