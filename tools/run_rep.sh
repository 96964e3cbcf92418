set -e -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 100 python tools/bench_stem.py 2>&1 | grep -v amdgpu | tail -2
timeout -k 10 200 python tools/ab_step.py overlap_wgrad=1 2>&1 | grep -v amdgpu
