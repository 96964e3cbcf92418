set -e -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 200 python tools/ab_step.py overlap_wgrad=1 2>&1 | grep -v amdgpu
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:30], d['roofline']['achieved'])"
