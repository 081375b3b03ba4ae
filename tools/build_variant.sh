#!/bin/bash
# Builds the library with extra compiler flags into ab/<name>.so (for tools/ab.sh); the in-tree build is left alone.
# usage: tools/build_variant.sh <name> [-DFLAG=... ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/ab"
cd "$ROOT/sift_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -fconstexpr-steps=200000000 -Wno-unused-value -Wno-unused-result "$@" -shared -o "$ROOT/ab/$NAME.so" sift_hip.hip
echo "built ab/$NAME.so ($*)"
