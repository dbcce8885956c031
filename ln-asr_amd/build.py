#!/usr/bin/env python3
"""Builds liblnasr_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python ln-asr_amd/build.py [--force] [--keep-temps] [--host-asan] [--variant NAME -DMACRO[=v] ...]

One object per csrc/*.hip (compiled in parallel, rebuilt only when stale), linked with RCCL
into ln-asr_amd/lnasr/liblnasr_hip.so -- in-tree, so the library travels with the snapshot.

--host-asan builds a second library, liblnasr_hip_asan.so, whose HOST side (the C-ABI entry points, argument
checks, staging, the upload pipeline's threads) is instrumented with AddressSanitizer + UBSan; the device code is
compiled as usual (-fno-gpu-sanitize: GPU ASan needs xnack+, which this pool does not run).  Load it with
LNASR_LIB=<path> and LD_PRELOAD=<asan_runtime()> (tests/test_host_asan.py).

--variant NAME with -D macros builds liblnasr_hip_NAME.so in its own object directory: experiment builds that are
measured against the shipped library in the same gpurun call (LNASR_LIB=<path> selects one); never the default.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lnasr", "liblnasr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=on", "-I/opt/rocm/include"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


ASAN_LIB = os.path.join(HERE, "lnasr", "liblnasr_hip_asan.so")
ASAN_FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-DLNASR_DEBUG_GUARDS", "-std=c++17", "-fPIC", "-fno-omit-frame-pointer",
              "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined",
              "-Wno-unused-function", "-Wno-unused-variable", "-ffp-contract=on", "-I/opt/rocm/include"]


def asan_runtime():
    """The shared AddressSanitizer runtime of the ROCm clang (to LD_PRELOAD into an uninstrumented python)."""
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


def build(force=False, keep_temps=False, verbose=True, host_asan=False, variant=None, defines=()):
    obj_dir = OBJ + "_asan" if host_asan else OBJ
    lib = ASAN_LIB if host_asan else LIB
    flags = ASAN_FLAGS if host_asan else FLAGS
    if variant:
        obj_dir = OBJ + "_" + variant
        lib = os.path.join(HERE, "lnasr", "liblnasr_hip_%s.so" % variant)
        flags = FLAGS + list(defines)
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    jobs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + flags + (["-save-temps=obj"] if keep_temps else []) + ["-c", s, "-o", o]
            jobs.append((s, cmd))

    def run(job):
        s, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    failed = False
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for s, rc, out in ex.map(run, jobs):
            if verbose or rc:
                print("[hipcc] %s %s" % (os.path.basename(s), "ok" if rc == 0 else "FAILED"))
            if out.strip() and (rc or verbose):
                print(out)
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        if host_asan:
            cmd += ["-fsanitize=address,undefined", "-shared-libsan", "-fno-gpu-sanitize"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print("[link] %s" % lib)
    return lib


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv, host_asan="--host-asan" in sys.argv,
          variant=_variant, defines=[a for a in sys.argv[1:] if a.startswith("-D")])
