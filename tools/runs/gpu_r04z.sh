#!/bin/bash
# round 4: can a process move its context out of the slow cross-attention state by re-creating it beside a placeholder allocation?
set -o pipefail
mkdir -p gpurun_out/r04z
for i in 1 2 3; do
timeout -k 10 300 python tools/es_place_probe.py 2048 3 80 bf16 > gpurun_out/r04z/probe_bf16_$i.txt 2>&1 || { tail -20 gpurun_out/r04z/probe_bf16_$i.txt; exit 1; }
echo "--- process $i (bf16)"; cat gpurun_out/r04z/probe_bf16_$i.txt
done
for i in 1 2; do
timeout -k 10 300 python tools/es_place_probe.py 2048 2 130 f16x3 > gpurun_out/r04z/probe_x3_$i.txt 2>&1 || { tail -20 gpurun_out/r04z/probe_x3_$i.txt; exit 1; }
echo "--- process $i (f16x3)"; cat gpurun_out/r04z/probe_x3_$i.txt
done
