"""GPU: parity of every gather-GEMM kernel form, each forced through mgd_conv_desc.form / form_arg (the library reads no
environment) in a process of its own: the thin-tile, producer/consumer and global-weight forms, the counted-pipeline kernel
with 4 waves at three pixel tiles, its ping-pong form with 128- and 192-pixel tiles, the phased 8-wave kernel of round 4 in
its three tile shapes, and the library's own dispatch - forward with statistics, folded inference epilogue and data gradient
with addend + fused BatchNorm-backward sums against fp32 torch (tests/gemm_form_check.py).  A geometry the forced form
refuses (MGD_EINVAL) is repeated under the library's dispatch and reported; the form must have run at least once."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

FORMS = [  # name, form, form_arg, expected kernel family
    ("library dispatch", 0, 0, "conv_gather_gemm"),
    ("producer / consumer", 6, 0, "producer/consumer"),
    ("global weight fragments, wave-uniform taps", 8, 0, "global weight fragments"),
    ("global weight fragments, per-lane taps (round 2)", 8, 1, "global weight fragments"),
    ("counted pipeline, 4 waves, cost-model tile", 9, 0, "counted pipeline"),
    ("counted pipeline, 4 waves, 192 pixels", 9, 12, "counted pipeline"),
    ("counted pipeline, 4 waves, 64 pixels", 9, 4, "counted pipeline"),
    ("ping-pong, 128 pixels", 10, 8, "ping-pong"),
    ("ping-pong, 192 pixels", 10, 12, "ping-pong"),
    ("phased 8 waves, 256 x 256", 12, 0, "phased"),
    ("phased 8 waves, 256 x 192", 12, 1, "phased"),
    ("phased 8 waves, 128 x 384", 12, 2, "phased"),
    # K order: form_arg + 256 = tap-major where chunk-major is the default, + 512 = chunk-major where tap-major is
    ("global weight fragments, tap-major K order", 8, 256, "global weight fragments"),
    ("counted pipeline, chunk-major K order", 9, 512, "counted pipeline"),
    ("ping-pong, 128 pixels, chunk-major K order", 10, 8 + 512, "ping-pong"),
    ("phased 8 waves, 256 x 192, chunk-major K order", 12, 1 + 512, "phased"),
]


@pytest.mark.parametrize("name,form,arg,expect", FORMS, ids=[f[0] for f in FORMS])
def test_gather_gemm_form_parity(name, form, arg, expect):
    e = dict(os.environ)
    for k in ("MGD_CONV_FORM", "MGD_CONV_FORM_ARG", "MGD_CONV_FORM_SOFT"):
        e.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(HERE, "gemm_form_check.py"), str(form), str(arg), expect], env=e,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, f"{name}:\n{out.stdout[-3000:]}\n{out.stderr[-2000:]}"
