#!/bin/bash
# A build of libminivideo.so in which ONE kernel file is compiled with other device flags (scheduling strategies and the like), the
# others as the product (minivideo_amd/build.py: extra_flags).  usage: tools/build_flag_variant.sh <name> <file base> "<flags>"
# -> abl_tmp/<name>/libminivideo.so ; the register check of the batch kernels runs on the ISA these flags produce.
set -e
NAME=$1; FILE=$2; FLAGS=$3
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/abl_tmp/$NAME
mkdir -p $O/t
python3 $R/minivideo_amd/build.py > /dev/null
INC="-I$R/include -I$R/minivideo_amd/csrc/hip -I$R/minivideo_amd/csrc/host"
OBJS=""
for s in $R/minivideo_amd/csrc/hip/*.hip; do
  b=$(basename $s .hip)
  if [ $b = $FILE ]; then
    mkdir -p $O/temps
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value $FLAGS -save-temps=obj -c $s -o $O/temps/$b.hip.o $INC 2>/dev/null
    case $b in recon_quad|recon_oct) python3 $R/tools/check_prefetch_hazard.py $O/temps/$b-hip-amdgcn-amd-amdhsa-gfx950.s > $O/$b.check 2>&1 || { echo "ISA CHECK FAILED for $NAME"; tail -2 $O/$b.check; rm -rf $O; exit 1; } ;; esac
    grep -A8 "name:.*${b}.*" $O/temps/$b-hip-amdgcn-amd-amdhsa-gfx950.s | grep -m2 "vgpr_count\|vgpr_spill" | tr '\n' ' '; echo
    cp $O/temps/$b.hip.o $O/t/$b.o; rm -rf $O/temps
  else
    cp $R/minivideo_amd/build/$b.hip.o $O/t/$b.o
  fi
  OBJS="$OBJS $O/t/$b.o"
done
HOST=$(ls $R/minivideo_amd/build/*.o | grep -v "\.hip\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 --hip-link -shared -fPIC -pthread -o $O/libminivideo.so $HOST $OBJS
rm -rf $O/t
echo built $O/libminivideo.so
