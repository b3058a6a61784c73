"""Build libjsim_mpc.so (HIP, gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.  Nine translation units in parallel
(one per horizon with a register kernel + the rest of the library): a minute instead of three on eight cores."""
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
               # MachineLICM hoists the materialisation of every 64-bit literal of the inlined sin / cos / tan polynomials
               # (and other loop-invariant address arithmetic) out of the K-tick loop, where the values then sit in
               # registers across the whole solve -- or, as happened, in scratch: ~20 doubles stored before the loop and
               # reloaded in every tick.  Without it: no scratch at T = 13 / 20 / 30 (was 0 / 12 / 152-208 B), 40-60 fewer registers, 0 B
               # (was 360 B) in the four-wave T = 40 kernel, speed within +-2 % on every configuration.  (The one kernel with
               # scratch today is T = 20's two-waves-per-SIMD form: 60 B, stored once per launch.)
               "-mllvm", "-disable-machine-licm"]
# (Rounds 1-2 also passed -mllvm -amdgpu-spill-vgpr-to-agpr=0, first because a wrong result was blamed on AGPR spills, then
# "because the evidence was collected with it".  Round 3 found the real cause of the wrong-row-id builds -- a live-range-split
# copy placed in front of a join block's exec restore, DESIGN.md section 5 fact 6 -- which that option has no bearing on; every
# build is now checked for that pattern (check_isa below) and the option is gone.)

OBJ_DIR = os.path.join(os.path.dirname(_HERE), "build", "obj")   # hipcc's -save-temps output (the device .s the guard reads)


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


def check_isa(asm_path: str) -> None:
    """The build-time guard of DESIGN.md section 5, fact 6: refuse a library in which a vector instruction sits in a join
    block in front of that block's exec restore (tools/isa_exec_check.py explains the pattern and how it miscomputes)."""
    import importlib.util
    tool = os.path.join(os.path.dirname(_HERE), "tools", "isa_exec_check.py")
    spec = importlib.util.spec_from_file_location("jsim_isa_exec_check", tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    findings = mod.check(asm_path)
    if findings:
        lines = [f"{k[:70]} block {b}: {len(ins)} vector instruction(s) in front of the exec restore at {asm_path}:{ln}"
                 for k, b, ln, ins in findings]
        raise RuntimeError("libjsim_mpc.so NOT installed -- the compiler placed vector code in front of a join block's exec "
                           "restore (lanes outside the mask keep stale values; DESIGN.md section 5, fact 6):\n  " + "\n  ".join(lines))


KERNEL_TUS = (13, 15, 16, 20, 25, 30, 32, 40)   # one translation unit per horizon with a register kernel (csrc/jsim_mpc.hip, JSIM_KERNEL_TU)
ASM_NAME = "jsim_mpc-hip-amdgcn-amd-amdhsa-gfx950.s"


def _build_split(hipcc: str, extra, obj_dir: str, tmp_lib: str, verbose: bool) -> None:
    """The library from nine translation units compiled in parallel: csrc/jsim_mpc.hip once per horizon (-DJSIM_KERNEL_TU=T: that
    horizon's register kernels, explicitly instantiated) and once for everything else (-DJSIM_SPLIT_BUILD: the same kernels declared
    `extern template`), then one link.  Every unit's device listing goes through the ISA guard; the listings are concatenated into
    the path the one-unit build leaves its listing at (tests and tools read that file)."""
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    units = [("main", ["-DJSIM_SPLIT_BUILD"])] + [(f"T{t}", [f"-DJSIM_KERNEL_TU={t}"]) for t in KERNEL_TUS]
    procs = []
    for name, defs in units:
        d = os.path.join(obj_dir, "tu_" + name)
        os.makedirs(d, exist_ok=True)
        cmd = [hipcc] + flags + extra + defs + ["-save-temps=obj", "-I", INC, "-c", SRC, "-o", os.path.join(d, name + ".o")]
        if verbose:
            print(" ".join(cmd))
        procs.append((name, d, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for name, d, cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"--- {name}: {' '.join(cmd)}\n{out[-4000:]}")
    if failed:
        raise RuntimeError("libjsim_mpc.so: compilation failed\n" + "\n".join(failed))
    for name, d, _, _ in procs:
        check_isa(os.path.join(d, ASM_NAME))
    with open(os.path.join(obj_dir, ASM_NAME), "w") as cat:
        for name, d, _, _ in procs:
            cat.write(open(os.path.join(d, ASM_NAME)).read())
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(d, name + ".o") for name, d, _, _ in procs] + ["-o", tmp_lib]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    # hipcc's other temporaries (bitcode, preprocessed sources, host listings: 140 MB) are of no use once the library is linked;
    # the concatenated device listing stays (tests/test_isa_guard.py and tools/isa_*.py read it)
    for name, d, _, _ in procs:
        shutil.rmtree(d, ignore_errors=True)
    for f in os.listdir(obj_dir):
        if f.startswith("jsim_mpc") and f != ASM_NAME:
            os.remove(os.path.join(obj_dir, f))


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        extra = os.environ.get("JSIM_HIPCC_EXTRA", "").split()   # diagnostic builds (-DJSIM_STAMPS, -DJSIM_DEV_ONLY_T40, ...)
        out = os.environ.get("JSIM_LIB_OUT", LIB)
        obj_dir = OBJ_DIR if out == LIB else os.path.join(os.path.dirname(os.path.abspath(out)), "obj")
        os.makedirs(obj_dir, exist_ok=True)
        tmp_lib = os.path.join(obj_dir, "libjsim_mpc.so")
        # one unit: the development builds that instantiate a subset (-DJSIM_DEV_*), or on request (JSIM_MONO_BUILD=1)
        mono = os.environ.get("JSIM_MONO_BUILD") == "1" or any(f.startswith("-DJSIM_DEV_") for f in extra)
        if mono:
            cmd = [_hipcc()] + HIPCC_FLAGS + extra + ["-save-temps=obj", "-I", INC, SRC, "-o", tmp_lib]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            check_isa(os.path.join(obj_dir, ASM_NAME))
        else:
            _build_split(_hipcc(), extra, obj_dir, tmp_lib, verbose)
        shutil.copyfile(tmp_lib, out)
        os.chmod(out, 0o755)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
