#!/bin/bash
# streaming stores for the saved-for-backward operand only (GELU' codes / pre-activation): epi_nt = 2
set -e
mkdir -p gpurun_out/r04_nt2
for rnd in 1 2 3; do
timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity > gpurun_out/r04_nt2/c2_nt0_r$rnd.json 2> gpurun_out/r04_nt2/c2_nt0.err
timeout -k 10 300 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --opt epi_nt=2 > gpurun_out/r04_nt2/c2_nt2_r$rnd.json 2> gpurun_out/r04_nt2/c2_nt2.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_nt2/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1]); k = j['kernels']
    print(f.split('/')[-1], j['ms_per_step'], j['value'], 'gemm_nt', k['gemm_nt']['ms_per_step'], 'attn', round(k['attn_fwd']['ms_per_step'] + k['attn_bwd']['ms_per_step'], 2), 'ln', round(k['layernorm_fwd']['ms_per_step'] + k['layernorm_bwd']['ms_per_step'], 2), 'wgrad', k['gemm_wgrad']['ms_per_step'])
PY
