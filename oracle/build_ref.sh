#!/usr/bin/env bash
# Build the REAL reference (DeadAt0m/ActiveSparseShifts-PyTorch, CPU + QuantizedCPU + Autograd +
# composite ops) from the sources where they lie under /root/reference into oracle/_ref/_C.so.
#
# TEST INFRASTRUCTURE ONLY -- used to (a) generate tests/golden/*.npz (tests/golden/make_golden.py),
# (b) cross-check oracle/shift_oracle.c, (c) serve as bench.py's cpu_baseline ("kind": "reference").
# Nothing from the reference is copied into this repository: g++ reads the sources in place and
# the only output is oracle/_ref/ (git-ignored, but shipped to the GPU box like our own .so files).
#
# The reference's own build system (setup.py) is NOT run.  Flags mirror it (setup.py:78-107):
# -std=c++17 -O3, -fopenmp -DAT_PARALLEL_OPENMP=1, -DTORCH18, no WITH_CUDA (CPU-only build).
#
# One translation unit needs a build-time patch on torch >= 2.x: quantized/shifts_quantized.cpp:126
# passes a std::string to AT_DISPATCH_QINT_TYPES, which now requires a `const char*`.  The file is
# streamed through sed into the compiler's stdin (never written anywhere); the patch replaces the
# `name` argument of that one macro call with a string literal.
set -euo pipefail

REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
SRC="$REF/torchshifts/csrc"

if [ ! -d "$SRC" ]; then
    echo "build_ref.sh: $SRC not present (GPU box?) -- keeping any prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$OUT/obj"

PY=${PYTHON:-python3}
TORCH_INC=$($PY - <<'EOF'
import torch.utils.cpp_extension as e
print(" ".join("-I" + p for p in e.include_paths() ))
EOF
)
TORCH_LIB=$($PY -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
PY_INC=$($PY -c "import sysconfig; print('-I' + sysconfig.get_paths()['include'])")

CXXFLAGS="-std=c++17 -O3 -fPIC -fopenmp -DAT_PARALLEL_OPENMP=1 -DTORCH18 -DTORCH_EXTENSION_NAME=_C \
 -Wno-unused-but-set-variable -Wno-unused-variable -Wno-sign-compare -Wno-unknown-pragmas -Wno-unused-function \
 -I$SRC $TORCH_INC $PY_INC"

compile() { # src obj
    if [ ! -f "$2" ] || [ "$1" -nt "$2" ]; then
        echo "  CXX $1"
        g++ $CXXFLAGS -c "$1" -o "$2"
    fi
}

compile "$SRC/torchshifts.cpp"                    "$OUT/obj/torchshifts.o" &
compile "$SRC/ops/shifts.cpp"                     "$OUT/obj/shifts.o" &
compile "$SRC/ops/autograd/shifts_autograd.cpp"   "$OUT/obj/shifts_autograd.o" &
compile "$SRC/ops/cpu/shifts_cpu.cpp"             "$OUT/obj/shifts_cpu.o" &

Q="$SRC/ops/quantized/shifts_quantized.cpp"
if [ ! -f "$OUT/obj/shifts_quantized.o" ] || [ "$Q" -nt "$OUT/obj/shifts_quantized.o" ]; then
    echo "  CXX $Q (streamed through sed: AT_DISPATCH_QINT_TYPES name -> literal)"
    sed 's/AT_DISPATCH_QINT_TYPES(input.scalar_type(), name,/AT_DISPATCH_QINT_TYPES(input.scalar_type(), "q_shiftnd_cpu",/' "$Q" \
      | g++ $CXXFLAGS -iquote "$SRC/ops/quantized" -x c++ -c - -o "$OUT/obj/shifts_quantized.o" &
fi
wait

g++ -shared -fopenmp -o "$OUT/_C.so" "$OUT"/obj/*.o -L"$TORCH_LIB" -ltorch -ltorch_cpu -lc10 \
    -Wl,-rpath,"$TORCH_LIB"
echo "built $OUT/_C.so"
