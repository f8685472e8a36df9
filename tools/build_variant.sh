#!/bin/bash
# A variant of the library with ONE source rebuilt under extra flags:
#   tools/build_variant.sh <name> <source.hip> <flags...>   ->  tools/_variants/lkamd_<name>.so
set -eu
NAME=$1; SRC=$2; shift 2
HERE=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $HERE/tools/_variants
OBJ=$HERE/tools/_variants/$(basename $SRC .hip)_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" \
    -c $HERE/lkpy_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls $HERE/lkpy_amd/csrc/_obj/*.o | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/tools/_variants/lkamd_$NAME.so $OBJ $OTHERS
echo $HERE/tools/_variants/lkamd_$NAME.so
