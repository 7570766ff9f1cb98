"""Compile the reference's exact-CPU Cython modules from where they lie into oracle/_ref/.

TEST INFRASTRUCTURE ONLY.  Outputs (generated C + extension .so) go to oracle/_ref/ which is
git-ignored; no reference source is copied into the repository.  Usage:
    python oracle/build_ref.py /root/reference
"""
import os
import subprocess
import sys
import sysconfig

import numpy as np

MODS = {
    "inverse_op_cython": "inf/layers/emerging/inverse_op_cython.pyx",
    "solve_parallel_mc": "inf/utils/fastflow_inverse/solve_parallel_mc.pyx",
}


def main(ref):
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_ref")
    os.makedirs(out, exist_ok=True)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    inc = sysconfig.get_paths()["include"]
    for name, rel in MODS.items():
        src = os.path.join(ref, rel)
        csrc = os.path.join(out, name + ".c")
        so = os.path.join(out, name + ext)
        if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src):
            continue
        subprocess.check_call(["cython", "-3", "-o", csrc, src])
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-w", "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
             "-I", inc, "-I", np.get_include(), csrc, "-o", so])
        os.remove(csrc)  # keep only the binary: generated C is derived reference source
        print("built", so)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
