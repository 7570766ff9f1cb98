"""Build libinvflow_hip.so (the C-ABI HIP library) for gfx950 with plain hipcc.

    python inverse-flow_amd/build.py [--force] [--verbose]

No torch headers are involved: the library is a plain extern "C" shared object
(include/invflow.h).  Objects are compiled per source file in parallel and cached by mtime.
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libinvflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-ffp-contract=fast"] + os.environ.get("HIPCC_EXTRA", "").split()


# per-file flags.  scan_duo.hip: the SLP vectoriser packs the chain wave's epilogue into v_pk_* instructions, which issue
# slower beside MFMAs than the scalar forms (-3 us per scan without it)
FILE_FLAGS = {"scan_duo.hip": ["-fno-slp-vectorize"]}


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False, stamps=False):
    """stamps=True: the development build with the in-kernel cycle stamps (tools/*stamps.py), kept apart from the product
    library: lib/libinvflow_hip_stamps.so"""
    global FLAGS, OBJDIR, LIB
    if stamps:
        FLAGS = FLAGS + ["-DIFL_STAMPS"]
        OBJDIR = os.path.join(HERE, "build", "stamps")
        LIB = os.path.join(LIBDIR, "libinvflow_hip_stamps.so")
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o) or hdr_time > os.path.getmtime(o):
            jobs.append([HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return cmd, r.returncode, r.stdout

    with cf.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for cmd, rc, out in ex.map(run, jobs):
            if out.strip() and (verbose or rc):
                print(out)
            if rc:
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if jobs or force or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        cmd, rc, out = run(cmd)
        if rc:
            print(out)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, stamps="--stamps" in sys.argv))
