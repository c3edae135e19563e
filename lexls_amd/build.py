"""In-tree build of the native library (hipcc -> gfx950).  Used by __graft_entry__.build()."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_native(force: bool = False) -> str:
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-s", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-s", "-C", csrc, "-j8"])
    path = os.path.join(csrc, "liblexls_hip.so")
    if not os.path.exists(path):
        raise RuntimeError("hipcc did not produce liblexls_hip.so")
    return path
