"""Build helpers: compile libcharon_hip.so (hipcc, gfx950) and the host front end in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcharon_hip.so")


def build(verbose=False):
    cmd = ["make", "-C", os.path.join(HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB):
        raise RuntimeError("libcharon_hip.so was not produced")
    return LIB
