set -e
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -2
timeout -k 10 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu | awk '{print $1,$2,$3,$4,$5,$6, "wgrad", $(NF-1), $NF}' | tail -34
