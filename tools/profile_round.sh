#!/bin/bash
# The rocprofv3 passes behind the published kernel figures (run on the GPU box from the repo root):
#   tools/profile_round.sh <out-dir under gpurun_out/>
# Kernel trace + stats of the default bench, HBM traffic (FETCH_SIZE, WRITE_SIZE: a pass each) and the SQ counters of the
# edge-stage kernels alone (tools/front_probe.py), and a kernel trace of a batch-512 run (two frames per CU in one launch).
out=gpurun_out/${1:-prof}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -x
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-streaming > $out/bench_under_rocprof.json 2> $out/bench.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/front_probe.py > $out/fetch.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/front_probe.py > $out/write.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $out/sq -- python3 tools/front_probe.py > $out/sq.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/b512 -- python3 bench.py --batch 512 --slots 3 --steps 12 --warmup 3 --no-cpu-baseline --no-streaming > $out/bench_b512_under_rocprof.json 2> $out/b512.err || exit 1
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench_plain.err || exit 1
echo done
