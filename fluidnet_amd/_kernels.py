"""Which source file each profiled kernel name of the library lives in (the names the built-in profiler and
tools/pmc_traffic.py report), and the git blob hash of a file: bench.py refuses a stored PMC traffic figure whose
kernel source has changed since it was measured."""
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")

KERNEL_SOURCE = {
    "k_conv3_in": "conv_mfma16.hip", "k_conv3_mid": "conv_mfma16.hip", "k_conv3_tail": "conv_mfma16.hip",
    "k_vel_fwd": "advect_vel3.hip+advect_vel3.inc+advect_vel3_kz1.inc", "k_vel_bwd": "advect_vel3.hip+advect_vel3.inc+advect_vel3_kz1.inc",
    "k_scalar_fwd": "advect_scalar3.hip", "k_scalar_bwd": "advect_scalar3.hip", "k_minmax3": "advect.hip",
    "k_curl": "vorticity.hip", "k_confine": "vorticity.hip", "k_vort_fused": "vorticity.hip", "k_stream_copy": "stencil.hip",
    "k_add_buoyancy": "stencil.hip", "k_add_gravity": "stencil.hip",
    "k_bcs_div_stats": "model.hip", "k_reduce_stats": "model.hip", "k_project": "model.hip", "k_net_input": "model.hip",
    "k_apply_bcs_indexed": "model.hip", "k_bc_scan": "model.hip",
}
# the default conv path can be switched by TFL_CONV_PATH; these are the files behind the same profiler names then
CONV_SOURCE = {"winograd": "conv_valu.hip", "mfma": "conv_mfma.hip", "direct": "conv.hip", "mfma16": "conv_mfma16.hip"}


def blob_sha(path):
    """`git hash-object <path>` without git"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def files_sha(names):
    """one hash over several files of csrc ("a+b+c": a kernel whose code is spread over includes): sha1 of their blob hashes"""
    parts = [blob_sha(os.path.join(CSRC, n)) for n in names.split("+")]
    return parts[0] if len(parts) == 1 else hashlib.sha1("".join(parts).encode()).hexdigest()


def source_of(kernel, conv_path="mfma16"):
    f = KERNEL_SOURCE.get(kernel)
    if f and kernel.startswith("k_conv3_"):
        f = CONV_SOURCE.get(conv_path, f)
    return f


def source_sha(kernel, conv_path="mfma16"):
    f = source_of(kernel, conv_path)
    if not f or not all(os.path.exists(os.path.join(CSRC, n)) for n in f.split("+")):
        return (f, None)
    return (f, files_sha(f))
