#!/bin/bash
# container: builds the library with extra compile-time flags into contrast_renderer_amd/build/variants/lib_<name>.so (travels with gpurun; the shipped
# library is rebuilt afterwards). usage: tools/build_variant.sh <name> "<flags>" [<file.hip> ...]   (files to recompile; default raster_edges.hip)
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
files=${@:-raster_edges.hip}
mkdir -p contrast_renderer_amd/build/variants
for f in $files; do touch contrast_renderer_amd/csrc/$f; done
CRH_EXTRA_FLAGS="$flags" python contrast_renderer_amd/build.py > /dev/null 2>&1
cp contrast_renderer_amd/libcontrast_hip.so contrast_renderer_amd/build/variants/lib_$name.so
for f in $files; do touch contrast_renderer_amd/csrc/$f; done
env -u CRH_FILE_FLAGS -u CRH_EXTRA_FLAGS python contrast_renderer_amd/build.py > /dev/null 2>&1  # (the tree's own library again, with the tree's own flags)
echo "built lib_$name.so ($flags)"
