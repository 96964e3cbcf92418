set -e
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "multiscale" 2>&1 | tail -3
for m in diou soft wbf; do timeout -k 10 200 python tools/bench_infer.py --method $m 2>&1 | grep -v amdgpu | tail -1; done
timeout -k 10 200 python tools/bench_infer.py --batch 1 --steps 100 2>&1 | grep -v amdgpu | tail -1
