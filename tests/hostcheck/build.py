"""Builds tests/hostcheck/_build/libhostcheck.so: the device arithmetic headers of
plonk_amd/csrc compiled for the host with g++ (test infrastructure)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhostcheck.so")
_CSRC = os.path.join(_HERE, "..", "..", "plonk_amd", "csrc")


def build():
    srcs = [os.path.join(_HERE, "hostcheck.cpp")] + [
        os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".cuh")
    ]
    newest = max(os.path.getmtime(s) for s in srcs)
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(
            ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", _SO,
             os.path.join(_HERE, "hostcheck.cpp")]
        )
    return _SO


def load():
    return ctypes.CDLL(build())
