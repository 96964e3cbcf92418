"""GPU: parity of every gather-GEMM kernel form built in round 3, each forced through the library's switches in a process
of its own (the switches are read once per process): the counted-pipeline kernel with 4 and 8 waves, its ping-pong form
with 128- and 192-pixel tiles, the streaming ping-pong kernel, and the default dispatch - forward with statistics, folded
inference epilogue and data gradient with addend against fp32 torch (tests/gemm_variant_check.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

VARIANTS = [
    ("default dispatch", {}, "conv_gather_gemm"),
    ("counted pipeline, 4 waves", {"MGD_GEMM9": "1", "MGD_GEMM9_WC": "4"}, "counted pipeline"),
    ("counted pipeline, 8 waves, 192 pixels", {"MGD_GEMM9": "1", "MGD_GEMM9_WC": "8", "MGD_GEMM9_NT": "12"}, "counted pipeline"),
    ("ping-pong, 128 pixels", {"MGD_GEMM9": "1", "MGD_GEMM9_WC": "8", "MGD_GEMM9_NT": "8", "MGD_GEMM9_PP": "1"}, "ping-pong"),
    ("ping-pong, 192 pixels", {"MGD_GEMM9": "1", "MGD_GEMM9_WC": "8", "MGD_GEMM9_NT": "12", "MGD_GEMM9_PP": "1"}, "ping-pong"),
    ("streaming ping-pong", {"MGD_GEMM10": "1"}, "streaming ping-pong"),
    ("round-2 kernel only", {"MGD_GEMM9": "0", "MGD_GEMM8_UNI": "0"}, "global weight fragments"),
]


@pytest.mark.parametrize("name,env,expect", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_gather_gemm_variant_parity(name, env, expect):
    e = dict(os.environ)
    for k in ("MGD_GEMM9", "MGD_GEMM9_WC", "MGD_GEMM9_NT", "MGD_GEMM9_PP", "MGD_GEMM10", "MGD_GEMM8_UNI", "MGD_DBG"):
        e.pop(k, None)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(HERE, "gemm_variant_check.py"), expect], env=e, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, f"{name}:\n{out.stdout[-3000:]}\n{out.stderr[-2000:]}"
