#!/bin/bash
# Build the extension of another revision (or of the working tree with extra flags) into ab/<name>.so for A/B timing
# with tools/k1_ab.py / k9_ab.py:   tools/build_rev.sh <git rev | WORK> <name> [extra hipcc flags]
set -e
rev=$1; name=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/ab"
if [ "$rev" = "WORK" ]; then
  src=$root
else
  src=$(mktemp -d /tmp/vkrev_XXXX)
  git -C "$root" archive "$rev" varkoder_amd/csrc include | tar -x -C "$src"
fi
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I "$src/include" "$src/varkoder_amd/csrc/vkimg.hip" -o "$root/ab/$name.so" "$@"
[ "$rev" = "WORK" ] || rm -rf "$src"
echo "$root/ab/$name.so"
