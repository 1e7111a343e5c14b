"""The verifier's host code (plonk_amd/csrc/verify.hip: Fq2/Fq6/Fq12 towers, Miller loop, final exponentiation,
G2 decoding; no GPU call in it) compiled for the host with AddressSanitizer and UndefinedBehaviorSanitizer and run
on the fifteen openings behind plonk/data/proof.json: the verdicts of kzg.rs:91-98, and no report from either
sanitizer (either one aborts the run)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from oracle import capi as oc
from tests.test_verify import openings, setup, tau_g2  # noqa: F401  (fixtures)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "hostcheck", "_build")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_verifier_host_code_under_sanitizers(openings, tau_g2, tmp_path):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "verify_sanitized")
    srcs = [os.path.join(ROOT, "plonk_amd", "csrc", "verify.hip"), os.path.join(ROOT, "tests", "hostcheck", "verify_sanitized.cpp")]
    headers = [os.path.join(ROOT, "plonk_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "plonk_amd", "csrc"))
               if f.endswith(".cuh")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(s) for s in srcs + headers):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-Wno-unknown-pragmas",
                               "-I" + os.path.join(ROOT, "include"), "-x", "c++", srcs[0], srcs[1], "-o", exe])
    coms = oc.g1_to_limbs([c[0] for c in openings])
    us = oc.fr_to_mont([c[1] for c in openings])
    vs = oc.fr_to_mont([c[2] for c in openings])
    proofs = oc.g1_to_limbs([c[3] for c in openings])
    vec = tmp_path / "openings.bin"
    with open(vec, "wb") as f:
        f.write(tau_g2)
        f.write(struct.pack("<Q", len(openings)))
        for a in (coms, us, vs, proofs):
            f.write(np.ascontiguousarray(a, dtype=np.uint64).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:verify_asan_link_order=0", UBSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([exe, str(vec)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-2000:]
    lines = run.stdout.strip().splitlines()
    assert [l for l in lines if l.startswith("single")] == ["single %d rc 0 ok 1" % i for i in range(len(openings))]
    assert "batch rc 0 ok 1" in lines and "tampered rc 0 ok 0" in lines and any(l.startswith("tau_g2 rc 0") for l in lines)
