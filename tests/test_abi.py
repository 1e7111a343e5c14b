"""The C-ABI library loads without a GPU and exports every symbol the headers under include/ declare."""
import os
import re

import plonk_amd
from plonk_amd import lib as plib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = "".join(open(os.path.join(ROOT, "include", name)).read() for name in ("plonk_hip.h", "plonk_hip_bench.h"))
    declared = set(re.findall(r"\b(plonk_hip_[a-z0-9_]+)\s*\(", header))
    surface = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    assert "plonk_hip_bench_" not in surface  # the hardware probes stay out of the drop-in surface
    assert declared, "no declarations found"
    lib = plonk_amd.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(plib.SIGNATURES)


def test_error_strings():
    lib = plonk_amd.load_library()
    assert lib.plonk_hip_strerror(0) == b"ok"
    # same text as the reference's CommitError::CommitFailed (plonk/src/common/kzg.rs:13)
    assert lib.plonk_hip_strerror(-1) == b"Length of tau_powers_g1 must be at least equal to degree of polynomial + 1"
    assert lib.plonk_hip_msm_phase_name(3) == b"accumulate"


def test_cpp_wrapper_compiles_as_plain_cpp(tmp_path):
    """include/plonk_hip.hpp needs nothing but the C header: a caller builds it with g++ alone."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use.cpp"
    src.write_text('#include "plonk_hip.hpp"\n'
                   "int main() { plonk_hip::DensePolynomial f(3); return (int)f.size() - 3; }\n")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"),
                           str(src)])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "kzg_roundtrip.cpp")])
