"""Builds libsbwt_amd.so and the `sbwt` CLI for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsbwt_amd.so")
BIN = os.path.join(HERE, "bin", "sbwt")
SOURCES = ["capi.cpp", "index_file.cpp", "host_pipeline.cpp", "device_index.hip", "map_kernels.hip", "builder.hip", "probe.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-value", "-pthread"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    headers.append(os.path.join(HERE, "..", "include", "sbwt_amd.h"))
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers):
            cmd = [hipcc] + FLAGS + os.environ.get("SBWT_CXXFLAGS", "").split() + ["-x", "hip", "-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-pthread"])
    main_src = os.path.join(CSRC, "sbwt_main.cpp")
    if force or _stale(BIN, [main_src, LIB]):
        subprocess.check_call([hipcc, "-O2", "-std=c++17", main_src, "-o", BIN, "-L" + os.path.dirname(LIB),
                               "-lsbwt_amd", "-Wl,-rpath,$ORIGIN/../lib", "-pthread"])
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
