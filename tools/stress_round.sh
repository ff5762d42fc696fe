#!/bin/bash
# Randomised stress runs of a round, on the GPU box:   tools/stress_round.sh r05
# (strict parity vs the oracle, tolerance and persistent contracts, slab decompositions in both arithmetics)
T=${1:-r05}
mkdir -p gpurun_out
python tools/stress_parity.py 2000 150000 gpurun_out/stress_${T}_parity.txt > /dev/null 2>&1
python tools/stress_parity.py 1500 160000 gpurun_out/stress_${T}_tolerance.txt tol > /dev/null 2>&1
python tools/stress_parity.py 3000 170000 gpurun_out/stress_${T}_persistent.txt persist > /dev/null 2>&1
python tools/stress_slab.py 800 190000 gpurun_out/stress_${T}_slab.txt > /dev/null 2>&1
ARITH=1 python tools/stress_slab.py 800 195000 gpurun_out/stress_${T}_slab_tolerance.txt > /dev/null 2>&1
tail -n 3 gpurun_out/stress_${T}_*.txt
