"""Builds libnerf_mi355x.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python nerf-projects_amd/build.py [--force] [--keep-temps]

hipcc cross-compiles without a GPU. The library links only libamdhip64: no torch types
cross the boundary (include/nerf_mi355x.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnerf_mi355x.so")
SOURCES = ["api.cpp", "train_api.cpp", "pack_weights.cpp", "mlp_kernel.hip", "ray_kernels.hip", "train_kernels.hip"]
HEADERS = [os.path.join(CSRC, "nerf_internal.h"), os.path.join(CSRC, "ctx_internal.h"),
           os.path.join(ROOT, "include", "nerf_mi355x.h")]
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
    "-ffp-contract=off",          # PyTorch's op boundaries are rounding boundaries; fmaf is explicit where wanted
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    # keep MFMA results in arch VGPRs where the allocator can: the per-layer ReLU then needs no
    # v_accvgpr_read per element (578 -> 163 in the MLP kernel, +1.1 % frame rate, measured A/B)
    "-mllvm", "-amdgpu-mfma-vgpr-form",
    "-x", "hip",
]


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
    cmd = [hipcc()] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    if keep_temps:
        tmp = os.path.join(HERE, "build")
        os.makedirs(tmp, exist_ok=True)
        cmd += ["-save-temps=cwd", "-Rpass-analysis=kernel-resource-usage"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=os.path.join(HERE, "build") if keep_temps else HERE)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print("built", LIB)
