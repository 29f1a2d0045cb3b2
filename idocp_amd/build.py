"""Build the native pieces in-tree.

  build_extension()  hipcc --offload-arch=gfx950  ->  idocp_amd/lib/libidocp_hip.so   (the product)
  build_oracle()     g++                          ->  oracle/liboracle.so, oracle/liboracle_hp.so (long double referee)   (test infrastructure)

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so files travel to the GPU box with the tree (they are git-ignored,
not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "idocp_amd", "csrc")
LIBDIR = os.path.join(ROOT, "idocp_amd", "lib")
OBJDIR = os.path.join(ROOT, "build", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
FLAGS += os.environ.get("IDOCP_EXTRA_HIPCC_FLAGS", "").split()      # diagnostic builds, e.g. -DIDOCP_S3_STAMPS (per-phase clock stamps of S3)


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(verbose=False, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(LIBDIR, "libidocp_hip.so")
    if force or jobs or not os.path.exists(lib):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build_oracle(verbose=False):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], capture_output=True, text=True)
    if verbose:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stderr)
    return os.path.join(ROOT, "oracle", "liboracle.so")


if __name__ == "__main__":
    print(build_extension(verbose=True, force="--force" in sys.argv))
    print(build_oracle(verbose=True))
