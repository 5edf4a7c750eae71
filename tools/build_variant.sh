#!/bin/bash
# Measurement builds of libminivideo.so with extra -D flags on the kernels (A/B timing only; results may differ from
# the product when a flag removes work).  usage: tools/build_variant.sh <name> -DFLAG [...]  -> abl_tmp/<name>/libminivideo.so
# run with MINIVIDEO_LIB=abl_tmp/<name>/libminivideo.so
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/abl_tmp/$NAME
mkdir -p $O
python3 $R/minivideo_amd/build.py > /dev/null
INC="-I$R/include -I$R/minivideo_amd/csrc/hip -I$R/minivideo_amd/csrc/host"
OBJS=""
for s in $R/minivideo_amd/csrc/hip/*.hip; do
  o=$O/$(basename $s).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value "$@" -c $s -o $o $INC
  OBJS="$OBJS $o"
done
HOST=$(ls $R/minivideo_amd/build/*.o | grep -v "\.hip\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 --hip-link -shared -fPIC -pthread -o $O/libminivideo.so $HOST $OBJS
echo built $O/libminivideo.so
