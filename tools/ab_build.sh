#!/bin/bash
# Build A/B variants of libbrx.so from different versions of brx_hot.S: tools/ab_build.sh name=<git rev or path> ...
# ("name=WORK" = the working tree).  Results: brotli-rs_amd/_ab/libbrx_<name>.so (travel to the GPU box; git-ignored).
set -eu
cd "$(dirname "$0")/.."
mkdir -p brotli-rs_amd/_ab
cp brotli-rs_amd/csrc/brx_hot.S /tmp/brx_hot_work.S
trap 'cp /tmp/brx_hot_work.S brotli-rs_amd/csrc/brx_hot.S' EXIT
for spec in "$@"; do
  name=${spec%%=*}; src=${spec#*=}
  if [ "$src" = WORK ]; then cp /tmp/brx_hot_work.S brotli-rs_amd/csrc/brx_hot.S
  elif [ -f "$src" ]; then cp "$src" brotli-rs_amd/csrc/brx_hot.S
  else git show "$src:brotli-rs_amd/csrc/brx_hot.S" > brotli-rs_amd/csrc/brx_hot.S; fi
  python brotli-rs_amd/build.py --force > /dev/null
  cp brotli-rs_amd/libbrx.so brotli-rs_amd/_ab/libbrx_$name.so
  echo "built $name"
done
cp /tmp/brx_hot_work.S brotli-rs_amd/csrc/brx_hot.S
python brotli-rs_amd/build.py --force > /dev/null
