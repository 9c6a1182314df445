#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA flags"  -> tools/bin/libmarl_NAME.so (a scratch copy of csrc is built;
# the product library and its build/ directory are not touched).  e.g. tools/build_variant.sh abl -DMARL_G3_ABLATE
set -euo pipefail
NAME=$1; EXTRA=${2:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /tmp/marl_variant.XXXXXX)
mkdir -p "$D/marlclassification_amd" "$ROOT/tools/bin"
cp -r "$ROOT/include" "$D/include"
mkdir -p "$D/marlclassification_amd/csrc"
cp "$ROOT"/marlclassification_amd/csrc/*.hip "$ROOT"/marlclassification_amd/csrc/*.h "$ROOT"/marlclassification_amd/csrc/Makefile "$D/marlclassification_amd/csrc/"
make -s -j8 -C "$D/marlclassification_amd/csrc" EXTRA="$EXTRA"
cp "$D/marlclassification_amd/csrc/libmarl_hip.so" "$ROOT/tools/bin/libmarl_$NAME.so"
rm -rf "$D"
echo "built tools/bin/libmarl_$NAME.so ($EXTRA)"
