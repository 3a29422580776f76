#!/usr/bin/env bash
# same-box A/B of variant libraries: tools/ab.sh "<kbench args>" lib1 lib2 ...
ARGS=$1; shift
for lib in "$@"; do
    echo "== $lib"
    SHIFTND_HIP_LIB=$PWD/variants/$lib.so python3 tools/kbench.py $ARGS 2>&1 | grep -E "fwd|bwd"
done
