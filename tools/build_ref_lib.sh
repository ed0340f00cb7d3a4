#!/bin/bash
# Developer tool: build libcdfo_hip.so of a git ref into cdfo_amd/lib/base/ (for same-box A/B runs through CDFO_LIB_PATH).
# usage: tools/build_ref_lib.sh [ref=HEAD]
set -e
REF=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$ROOT" archive "$REF" cdfo_amd/csrc include | tar -x -C "$T"
mkdir -p "$T/obj" "$ROOT/cdfo_amd/lib/base"
ls "$T"/cdfo_amd/csrc/*.hip | xargs -P 6 -I{} sh -c '/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=fast -Wno-unused-function -c {} -o '"$T"'/obj/$(basename {} .hip).o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/cdfo_amd/lib/base/libcdfo_hip.so" "$T"/obj/*.o
rm -rf "$T"
ls -la "$ROOT/cdfo_amd/lib/base/libcdfo_hip.so"
