#!/bin/bash
# Builds the working tree with extra -D flags into tools/ab/libmgx_<name>.so
#   tools/build_variant.sh <name> [-DX=1 ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cp -r "$root/fast-genomic-data-processing_amd" "$root/include" "$tmp/"
find "$tmp" -name "*.o" -delete
sed -i "s|\"-Wall\"|\"-Wall\", $(printf '"%s", ' "$@" | sed 's/, $//')|" "$tmp/fast-genomic-data-processing_amd/build.py"
python "$tmp/fast-genomic-data-processing_amd/build.py" --force > /dev/null
mkdir -p "$root/tools/ab"
cp "$tmp/fast-genomic-data-processing_amd/libmgx.so" "$root/tools/ab/libmgx_$name.so"
rm -rf "$tmp"
echo "built $name"
