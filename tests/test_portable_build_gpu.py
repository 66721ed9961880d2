"""`make portable` (csrc/Makefile: -DDIB_PORTABLE_TAPS) replaces every hand-written, register-naming loop of csrc/dib_blur.hip by
the C++ restatement that sits next to it -- the escape hatch for a compiler the asm was not tuned on (tests/test_kernel_resources.py
goes red there).  The restatements must BE the asm's arithmetic: the same launches through both builds, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PORTABLE = os.path.join(ROOT, "detectinblur_amd", "libdib_hip_portable.so")
CHILD = os.path.join(ROOT, "tests", "_portable_child.py")


def _digests(lib):
    env = dict(os.environ)
    env.pop("DIB_HIP_LIB", None)
    if lib:
        env["DIB_HIP_LIB"] = lib
    p = subprocess.run([sys.executable, CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("DIGESTS ")]
    assert line, p.stdout[-2000:]
    return json.loads(line[-1][8:])


@pytest.mark.gpu
def test_portable_build_equals_the_hand_written_loops_bit_for_bit():
    if not os.path.isfile(PORTABLE):
        p = subprocess.run(["make", "-C", os.path.join(ROOT, "detectinblur_amd", "csrc"), "portable"], capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
    product, portable = _digests(None), _digests(PORTABLE)
    assert set(product) == set(portable) and len(product) >= 16
    differing = sorted(k for k in product if product[k] != portable[k])
    assert not differing, differing
