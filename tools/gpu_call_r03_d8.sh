O=gpurun_out/r03_d8; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm_nt" > $O/test_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -4 $O/test_gemm.log
python -m pytest tests/test_gpu_configs.py tests/test_gpu_models.py -m gpu -q > $O/test_models.log 2>&1; echo "model tests rc=$?"; tail -4 $O/test_models.log
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_d8.json 2> $O/bench_d8.err; echo "bench d8 rc=$?"
CLIPK_GELU_AUX=bf16 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_bf16aux.json 2> $O/bench_bf16aux.err; echo "bench bf16 rc=$?"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_d8_b.json 2> $O/bench_d8_b.err
python3 - <<'P'
import json
for f in ("bench_d8","bench_bf16aux","bench_d8_b"):
    d=json.load(open(f"gpurun_out/r03_d8/{f}.json"))
    k=d["kernels"]
    print(f, d["ms_per_step"], d["value"], "gemm_nt", k["gemm_nt"]["ms_per_step"], "frac", d["roofline"]["frac"], d.get("parity",{}).get("loss_abs_err"))
    for m,v in d["roofline"]["by_epilogue"].items(): print("   ", m, v["avg_us"], v["frac"])
P
