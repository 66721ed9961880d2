"""The shipped kernel-choice data against the stack it runs on (detectinblur_amd/kernel_choices.py).

The find-db / perf-db files are named after ONE MIOpen build and the TunableOp file carries validator lines of one PyTorch /
hipBLASLt / rocBLAS build: on any other stack both are ignored SILENTLY -- the train step is then ~6 % slower and a fresh process
searches for a minute and a half.  These tests make that loud."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import detectinblur_amd
assert "MIOPEN_USER_DB_PATH" not in os.environ or os.environ.get("DIB_KERNEL_CHOICES_OWNER")      # importing exports nothing
detectinblur_amd.use_shipped_kernel_choices()
import torch
x = torch.randn(2, 64, 200, 336, device="cuda").to(memory_format=torch.channels_last)           # a shape of the shipped find-db (C2 of 800 x 1344)
conv = torch.nn.Conv2d(64, 64, 3, padding=1).cuda().to(memory_format=torch.channels_last)
t0 = time.perf_counter(); y = conv(x); torch.cuda.synchronize(); first = time.perf_counter() - t0
a = torch.randn(1000, 1024, device="cuda"); lin = torch.nn.Linear(1024, 91).cuda(); lin(a); torch.cuda.synchronize()   # a box-head GEMM
print("REPORT " + json.dumps(dict(detectinblur_amd.kernel_choices_report(), first_conv_s=first)))
''' % ROOT


def test_shipped_choices_belong_to_the_running_miopen_and_pytorch():
    env = {k: v for k, v in os.environ.items() if k not in ("MIOPEN_USER_DB_PATH", "DIB_KERNEL_CHOICES_OWNER") and not k.startswith("PYTORCH_TUNABLEOP")}
    p = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("REPORT ")][-1][7:])
    assert r["installed"] and r["miopen_user_db"]
    # MIOpen wrote under the shipped files' names: they belong to this build.  A foreign name here means the shipped data is dead
    # weight on this stack: regenerate it (`python -m detectinblur_amd.kernel_choices --fill --shapes all --install`; the report says so itself).
    assert r["miopen_foreign_files"] == [], r
    assert r["tunableop_validators_match"] is True, r
    assert r["tunableop_entries_loaded"] >= r["tunableop_shipped_entries"] >= 60, r
