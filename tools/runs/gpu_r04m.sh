#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04m
timeout -k 10 900 python tools/es_pitch_probe.py 2048 5 bf16 > gpurun_out/r04m/es_pitch_probe_bf16.txt 2>&1 || { tail -5 gpurun_out/r04m/es_pitch_probe_bf16.txt; exit 1; }
cat gpurun_out/r04m/es_pitch_probe_bf16.txt
