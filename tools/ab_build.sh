#!/bin/bash
# Build libx266hip.so of another git revision next to the working tree's, for same-box A/B timing:
#   tools/ab_build.sh <git-ref>      ->  tools/_ab/libx266hip_ref.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REF=${1:-HEAD}
T=$(mktemp -d)
git -C "$R" archive "$REF" x266_amd/csrc include | tar -x -C "$T"
make -C "$T/x266_amd/csrc" --no-print-directory >/dev/null
mkdir -p "$R/tools/_ab"
cp "$T/x266_amd/libx266hip.so" "$R/tools/_ab/libx266hip_ref.so"
rm -rf "$T"
echo "built $REF -> tools/_ab/libx266hip_ref.so"
