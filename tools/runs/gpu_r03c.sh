#!/bin/bash
# round 3, third GPU call: encoder-only CU mask (token loop unmasked), MX kernels at whisper-large-v3 widths, log-mel with 8 frames per workgroup
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
timeout -k 10 120 ./tools/mx_gemm_check > $O/mx_gemm_check.txt 2>&1; echo "mx_gemm_check rc $?"; cat $O/mx_gemm_check.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fp8 or mx" > $O/pytest_fp8.log 2>&1; echo "fp8 tests rc $?"; tail -12 $O/pytest_fp8.log
WH_MEL_FRB=8 timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "log_mel or golden_vectors or batch_equals_staged or longform" > $O/pytest_mel8.log 2>&1; echo "mel FRB=8 tests rc $?"; tail -4 $O/pytest_mel8.log
for frb in 16 8; do
  WH_MEL_FRB=$frb timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_mel$frb.json 2> $O/bench_mel$frb.err; echo "bench mel FRB=$frb rc $?"
  python3 -c "import json;j=json.load(open('$O/bench_mel$frb.json'));print(j['value'],j['ms_per_step'],j['kernel_group_ms_per_step'])"
done
for e in 32 48 64 80; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch1 --no-row-check --pipeline 1 --enc-cus $e --dec-cus 256 > $O/bench_pipe_e${e}_dall.json 2> $O/bench_pipe_e${e}_dall.err; echo "pipe enc $e / dec all rc $?"
  python3 -c "import json;j=json.load(open('$O/bench_pipe_e${e}_dall.json'));print(j['value'],j['ms_per_step'],j['stage_ms_per_step'],j['roofline']['frac'],j['kernel_group_ms_per_step'])"
done
timeout -k 10 100 ./tools/cu_mask_probe > $O/cu_mask_probe.txt 2>&1; echo "probe rc $?"; grep -v "^    xcc" $O/cu_mask_probe.txt | tail -25
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "bit_identical or largest_batch_rows or teacher_forced_agreement or batched_contexts" > $O/pytest_tile.log 2>&1; echo "tile kernel tests rc $?"; tail -6 $O/pytest_tile.log
for w in -3 -1; do
  WH_DEC_WIDE=$w timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-row-check > $O/bench_decwide$w.json 2> $O/bench_decwide$w.err; echo "bench WH_DEC_WIDE=$w rc $?"
  python3 -c "import json;j=json.load(open('$O/bench_decwide$w.json'));print(j['value'],j['ms_per_step'],j['stage_ms_per_step'],j['kernel_group_ms_per_step'])"
done
