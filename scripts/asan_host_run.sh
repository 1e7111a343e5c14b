#!/bin/bash
# The compiled example (examples/kzg_roundtrip.cpp) on a library whose HOST code is built with AddressSanitizer and
# UndefinedBehaviorSanitizer (device code as usual), next to the normal build on the same input: run on the GPU box
# from the repo root after `make -C plonk_amd/csrc asan` (which builds plonk_amd/_lib_asan).  usage: bash scripts/asan_host_run.sh DEGREE
set -u
DEG=${1:-4095}
OUT=gpurun_out
python3 - "$DEG" <<'PY'
import sys, numpy as np
n = int(sys.argv[1]) + 1
sc = np.frombuffer(np.random.default_rng(7).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
sc.tofile("gpurun_out/asan_coeffs.bin")
PY
ARGS="$DEG 123456789abcdef fedcba987654321 1122334455667788 0102030405060708 aaaa bbbb cccc 0dd"
timeout -k 10 120 plonk_amd/_lib/kzg_roundtrip $ARGS < $OUT/asan_coeffs.bin > $OUT/asan_plain.txt 2> $OUT/asan_plain.err || { echo "plain run failed"; exit 1; }
ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    timeout -k 10 300 plonk_amd/_lib_asan/kzg_roundtrip $ARGS < $OUT/asan_coeffs.bin > $OUT/asan_san.txt 2> $OUT/asan_san.err
echo "sanitized run rc $?"
cmp $OUT/asan_plain.txt $OUT/asan_san.txt && echo "same output"
grep -c "ERROR: AddressSanitizer\|runtime error" $OUT/asan_san.err
tail -n 4 $OUT/asan_san.txt
rm -f $OUT/asan_coeffs.bin
