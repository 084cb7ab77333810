"""Builds libnerf_mi355x.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python nerf-projects_amd/build.py [--force] [--keep-temps]

hipcc cross-compiles without a GPU. The library links only libamdhip64: no torch types
cross the boundary (include/nerf_mi355x.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnerf_mi355x.so")
SOURCES = ["api.cpp", "train_api.cpp", "pack_weights.cpp", "mlp_kernel.hip", "mlp_kernel_h2.hip", "mlp_bwd_kernel_h2.hip",
           "ray_kernels.hip", "train_kernels.hip", "train_dw_kernel.hip", "refresh_kernels.hip"]
HEADERS = [os.path.join(CSRC, "nerf_internal.h"), os.path.join(CSRC, "ctx_internal.h"),
           os.path.join(CSRC, "mlp_inputs.h"), os.path.join(CSRC, "mlp_pair_common.h"), os.path.join(CSRC, "ray_device.h"),
           os.path.join(ROOT, "include", "nerf_mi355x.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
    "-ffp-contract=off",          # PyTorch's op boundaries are rounding boundaries; fmaf is explicit where wanted
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    "-x", "hip",
]
# keep MFMA results in arch VGPRs where the allocator can: the per-layer ReLU then needs no
# v_accvgpr_read per element (578 -> 163 in the fp32 MLP kernel, +1.1 % frame rate, measured A/B).
# The fp16-pair kernel holds 128 accumulators + 176 operand registers and needs the AGPR half for the former.
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# train_dw_kernel.hip: 256 accumulators per lane, likewise.
EXTRA = {"mlp_kernel_h2.hip": [], "mlp_bwd_kernel_h2.hip": [], "train_dw_kernel.hip": []}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, keep_temps=False, verbose=True):
    if not force and up_to_date():
        return LIB
    tmp = os.path.join(HERE, "build")
    os.makedirs(tmp, exist_ok=True)
    only = os.environ.get("NERF_BUILD_ONLY")      # e.g. "mlp_kernel_h2.hip": recompile just that object

    def compile_one(src):
        obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
        if only and src not in only.split(",") and os.path.exists(obj):
            return obj
        cmd = [hipcc()] + FLAGS + EXTRA.get(src, VGPR_FORM) + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
        cmd += os.environ.get("NERF_EXTRA_FLAGS", "").split()       # ablation builds (-DNERF_ABLATE_...)
        if keep_temps:
            cmd += ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"]
        cmd += ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=tmp)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    out = os.environ.get("NERF_LIB_OUT", LIB)                       # ablation builds go next to the real library
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out + ".tmp"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True, cwd=tmp)
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print("built", LIB)
