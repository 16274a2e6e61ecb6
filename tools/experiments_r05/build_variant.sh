#!/bin/bash
# usage: tools/experiments_r05/build_variant.sh <name> [EXTRA flags...]  -> tools/experiments_r05/lib/libsdrhip_<name>.so
# (a variant of the product library for A / B runs on the GPU box: SDRHIP_LIB_PATH=<that file> python tools/...)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
make -C $ROOT/sdrdaemon_amd/csrc -j8 -s BUILD=/tmp/sdrhip_build_$NAME OUT=$ROOT/tools/experiments_r05/lib/libsdrhip_$NAME.so EXTRA="$*"
echo "built tools/experiments_r05/lib/libsdrhip_$NAME.so  (EXTRA: $*)"
