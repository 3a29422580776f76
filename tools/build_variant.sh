#!/usr/bin/env bash
# Build a variant of libshiftnd_hip.so for same-box A/B timing (tools/kbench.py with SHIFTND_HIP_LIB=...):
#   tools/build_variant.sh <name> <file.hip> [extra hipcc flags for that file...]
# recompiles one source with extra flags and links it with the in-tree objects into variants/<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/activesparseshifts-pytorch_amd
NAME=$1; SRC=$2; shift 2
mkdir -p $ROOT/variants
OBJ=$ROOT/variants/$NAME.$SRC.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I$ROOT/include -I$PKG/csrc "$@" -c $PKG/csrc/$SRC -o $OBJ
OBJS=""
for f in $PKG/build/*.hip.o; do
    [ "$(basename $f)" = "$SRC.o" ] && OBJS="$OBJS $OBJ" || OBJS="$OBJS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $ROOT/variants/$NAME.so
rm -f $OBJ
echo built variants/$NAME.so
