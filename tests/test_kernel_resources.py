"""Register and scratch budget of the blur's hot kernels, read from hipcc's own resource report of csrc/dib_blur.hip.

Why a test: these kernels are written against hard limits of gfx950 that nothing else enforces -- eight workgroups per CU need
<= 64 vector registers per lane and <= 80 scalar registers per wave (81..96 silently run seven waves per SIMD:
MI355X_MICROARCH.md, "Residency and cooperative launch"), and ANY scratch memory (a spilled register) adds a scratch set-up to
every workgroup's dispatch.  An innocent edit of the tile function, or of the compaction code that shares the step kernel with
it, crosses those limits without a single wrong pixel."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "detectinblur_amd", "csrc", "dib_blur.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# mangled-name fragment -> (max VGPRs, max SGPRs, waves per SIMD)
BUDGET = {
    "blur_quad_f16_kernelILi0ELi128ELb0E": (64, 80, 8),  # bit-exact, the blur alone
    "blur_quad_f16_kernelILi2ELi128ELb0E": (64, 80, 8),  # FMA16
    "blur_quad_f16_kernelILi0ELi128ELb1E": (64, 80, 8),  # the same two on the 1-D grid of a ragged batch
    "blur_quad_f16_kernelILi2ELi128ELb1E": (64, 80, 8),
    "blur_quad_f16_kernelILi3ELi128ELb0E": (64, 80, 8),  # FAST16 (vertical-run groups)
    "blur_quad_f16_kernelILi3ELi128ELb1E": (64, 80, 8),
    "blur_quad_f32acc_kernelILi128E": (64, 80, 8),       # DIB_ACC_FP32 on the default tiles
    "blur_step_f16_kernelILi0E": (64, 80, 8),            # the step's single launch: compaction + blur
    "blur_step_f16_kernelILi2E": (64, 80, 8),
}


@pytest.fixture(scope="module")
def report():
    if not os.path.isfile(HIPCC):
        pytest.skip("hipcc not available")
    p = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-c", SRC,
                        "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.mark.parametrize("kernel", list(BUDGET))
def test_hot_kernels_stay_inside_their_register_budget_without_scratch(report, kernel):
    names = [n for n in report if kernel in n]
    assert len(names) == 1, names
    r = report[names[0]]
    vgpr, sgpr, waves = BUDGET[kernel]
    assert r["ScratchSize"] == 0, r
    assert r["VGPRs"] <= vgpr and r["TotalSGPRs"] <= sgpr and r["Occupancy"] >= waves, r
