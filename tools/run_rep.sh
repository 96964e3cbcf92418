set -e
MGD_ILV=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv or fused" 2>&1 | tail -2
MGD_ILV=1 timeout -k 10 200 python tools/bench_conv.py 2>&1 | grep -v amdgpu | sed -n 1,16p
