#!/bin/bash
# Further counter passes (a run each; rocprofv3 does not split counters over passes):
#   tools/pmc_more.sh <out-dir under gpurun_out/>
# a, b, c: the front kernel alone (tools/front_probe.py) -- occupancy, memory-instruction cycles and levels, L1 stalls;
# d: every kernel of the bench loop (kernels are serialised under --pmc, so the counts are per kernel, the times exclusive).
out=gpurun_out/${1:-pmc_more}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -x
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $out/a -- python3 tools/front_probe.py > $out/a.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $out/b -- python3 tools/front_probe.py > $out/b.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY --output-format csv -d $out/c -- python3 tools/front_probe.py > $out/c.txt 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $out/d -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-streaming > $out/d.txt 2>&1 || exit 1
echo done
