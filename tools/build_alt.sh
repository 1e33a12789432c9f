#!/bin/bash
# Builds the tree's current sources as an alternative library robigo_luculenta_amd/librl_alt_<name>.so for tools/ab3.sh, from a
# snapshot under /tmp so that the tree can be edited while it compiles.  Usage: tools/build_alt.sh name [EXTRA flags...]
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SNAP=/tmp/rl_alt/$NAME
rm -rf $SNAP; mkdir -p $SNAP/robigo_luculenta_amd $SNAP/include
cp -r $ROOT/robigo_luculenta_amd/csrc $SNAP/robigo_luculenta_amd/
cp $ROOT/include/*.h $SNAP/include/
make -s -C $SNAP/robigo_luculenta_amd/csrc OUT=$ROOT/robigo_luculenta_amd/librl_alt_$NAME.so EXTRA="$*"
echo "built librl_alt_$NAME.so"
