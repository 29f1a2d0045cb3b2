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


def _digest(paths, extra=""):
    """sha256 over the CONTENTS of the given files (+ the flags): what an object was built from.  Modification times say nothing after a
    fresh clone or a copy to another box (every file is 'new'), and a stale object next to an edited header is the worse failure."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(stamp, digest):
    try:
        return open(stamp).read().strip() == digest
    except OSError:
        return False


def build_extension(verbose=False, force=False):
    """Compile every source whose CONTENT (or that of any header, or the flags) differs from what its object was built from, then link.
    A clean clone builds everything (34 s on 8 cores); an unchanged tree builds nothing; the library's own stamp ties it to its objects."""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    hdr_digest = _digest(hdrs, " ".join(FLAGS))
    objs = []
    jobs = []
    stamps = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s) + ".o")
        objs.append(o)
        d = _digest([s], hdr_digest)
        stamps.append(d)
        if force or not os.path.exists(o) or not _stamp_ok(o + ".sha256", d):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
            jobs.append((cmd, o + ".sha256", d))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r

    def compile_one(job):
        cmd, stamp, d = job
        if os.path.exists(stamp):
            os.remove(stamp)
        run(cmd)
        with open(stamp, "w") as f:
            f.write(d + "\n")

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    lib = os.path.join(LIBDIR, "libidocp_hip.so")
    lib_digest = _digest([], "".join(stamps))
    # the library's stamp lives NEXT to it and travels with it (a GPU box receives the .so without build/): it names the sources the
    # library was linked from, so a box that has the sources but a library from other sources rebuilds instead of running stale code
    if force or jobs or not os.path.exists(lib) or not _stamp_ok(lib + ".sha256", lib_digest):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        with open(lib + ".sha256", "w") as f:
            f.write(lib_digest + "\n")
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
