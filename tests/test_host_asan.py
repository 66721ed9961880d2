"""The hand-written host C code (csrc/host/dib_host.c: MT19937, polar gauss, trajectory walk, PSF splatting, centring, COCO matching, polygon masks) under
AddressSanitizer + UndefinedBehaviorSanitizer: `make -C detectinblur_amd/csrc asan` builds libdib_host_asan.so, and the
host-side parity tests run against it in a child interpreter with libasan preloaded.  Any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_library_is_clean_under_asan_and_ubsan():
    asan_rt = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isfile(asan_rt):
        pytest.skip("no libasan runtime on this machine")
    subprocess.run(["make", "-C", os.path.join(ROOT, "detectinblur_amd", "csrc"), "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "detectinblur_amd", "libdib_host_asan.so")
    assert os.path.isfile(lib)
    env = dict(os.environ, LD_PRELOAD=asan_rt, DIB_HOST_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host_native.py"),
                        os.path.join(ROOT, "tests", "test_transforms_cpu.py"),
                        os.path.join(ROOT, "tests", "test_coco_eval.py") + "::test_native_matching_equals_the_interpreted_loop_nest",
                        os.path.join(ROOT, "tests", "test_coco_eval.py") + "::test_native_accumulate_equals_the_interpreted_form",
                        os.path.join(ROOT, "tests", "test_masks.py") + "::test_object_masks_equal_the_reference",
                        os.path.join(ROOT, "tests", "test_masks.py") + "::test_random_polygons_against_the_reference_library",
                        os.path.join(ROOT, "tests", "test_masks.py") + "::test_convert_coco_polys_to_mask_target"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout[-3000:] + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail
    # the child really ran the sanitizer build
    probe = subprocess.run([sys.executable, "-c", "from detectinblur_amd import _hostlib; _hostlib.lib(); print(_hostlib.LIB_PATH)"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libdib_host_asan.so"), probe.stderr[-2000:]
