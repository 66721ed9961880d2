"""detectinblur_amd: MI355X-native implementation of detectInBlur's `--gpu_blur` hot path.

Module names mirror the reference (mohammed-amr/detectInBlur) so call sites read the same:
    detectinblur_amd.models.blur_functions.blur_image_list / manual_blur
    detectinblur_amd.utils.expand_targets / fix_bounding_box_squeeze / get_norm_params
    detectinblur_amd.transforms.BlurImage, detectinblur_amd.motion_blur.*
All device work goes through the C ABI of include/dib.h (libdib_hip.so, hand-written HIP for
gfx950); there is no CPU fallback.
"""
__version__ = "0.1.0"

# Shipped kernel-choice data for MIOpen / TunableOp is an explicit opt-in now (round-4 review: importing a drop-in library must
# not export MIOPEN_* / PYTORCH_TUNABLEOP_* into its caller's environment): the drivers and bench.py call
# `detectinblur_amd.use_shipped_kernel_choices()` before their first convolution; see kernel_choices.py.
# (resolved on first use, so that `python -m detectinblur_amd.kernel_choices` runs ONE copy of that module)


def __getattr__(name):
    if name in ("kernel_choices_report", "use_shipped_kernel_choices"):
        from . import kernel_choices
        return kernel_choices.report if name == "kernel_choices_report" else kernel_choices.use_shipped_kernel_choices
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
