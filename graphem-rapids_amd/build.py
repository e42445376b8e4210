"""In-tree build of libgraphem_hip.so (hipcc, gfx950).  `python graphem-rapids_amd/build.py`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgraphem_hip.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp"))]
    srcs.append(os.path.join(HERE, "..", "include", "graphem_hip.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libgraphem_hip.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
