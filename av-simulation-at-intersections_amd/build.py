"""Build libjsim_mpc.so (HIP, gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "jsim_mpc.hip")
INC = os.path.join(os.path.dirname(_HERE), "include")
LIB = os.path.join(_HERE, "libjsim_mpc.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off",  # S1-S3 follow numpy's operation order; fma() is explicit where wanted
               "-fno-fast-math",
               # Spills of these 400-500-register kernels go to scratch, not to "free" AGPRs: with the default
               # (spill-vgpr-to-agpr on) one instantiation of the two-wave kernel was miscompiled -- correct Hessian and
               # gradient, wrong active-set iterations, right again with this option, with machine sinking disabled, or
               # with any instrumentation of the loop.  Register counts and speed are the same either way.
               "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libjsim_mpc.so)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    csrc = os.path.join(_HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc"))] + [os.path.join(INC, "jsim_mpc.h")]
    return os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        extra = os.environ.get("JSIM_HIPCC_EXTRA", "").split()   # diagnostic builds (-DJSIM_STAMPS, -DJSIM_DEV_ONLY_T40, ...)
        cmd = [_hipcc()] + HIPCC_FLAGS + extra + ["-I", INC, SRC, "-o", os.environ.get("JSIM_LIB_OUT", LIB)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
