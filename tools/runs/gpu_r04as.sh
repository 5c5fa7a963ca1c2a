#!/bin/bash
# round 4: k_dec_cross_attn_es8 as two workgroups per CU (rings of four tiles, queries from global memory): check, then the fp8 bench line A/B in one call
set -o pipefail
mkdir -p gpurun_out/r04as
WH_ES8_FORM=2 timeout -k 10 240 ./tools/es8_check > gpurun_out/r04as/es8_check_form2.txt 2>&1 || { cat gpurun_out/r04as/es8_check_form2.txt; exit 1; }
cat gpurun_out/r04as/es8_check_form2.txt | cut -c1-200
for f in 1 2 1 2; do
WH_ES8_FORM=$f timeout -k 10 500 python bench.py --precision fp8 --clips 2048 --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 > gpurun_out/r04as/bench_fp8_form$f.json 2> gpurun_out/r04as/bench_fp8_form$f.err || { tail -20 gpurun_out/r04as/bench_fp8_form$f.err; exit 1; }
python - <<P
import json
d=json.load(open('gpurun_out/r04as/bench_fp8_form$f.json'))
print('form=$f', round(d['value']), round(d['ms_per_step'],1), d['kernel_group_ms_per_step']['dec_cross_attn'], round(d['roofline']['frac'],3), d['workspace_placement'])
P
done
