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
  b=$(basename $s .hip)
  mkdir -p $O/temps_$b
  X=""; case $b in recon_oct|recon_pipe) X="-Xarch_device -mllvm=-amdgpu-sched-strategy=max-ilp";; esac   # (as minivideo_amd/build.py: extra_flags)
  [ -n "$MVHP_NO_FILE_FLAGS" ] && X=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value $X "$@" -save-temps=obj -c $s -o $O/temps_$b/$b.hip.o $INC
  cp $O/temps_$b/$b.hip.o $o
  # the same ISA check as the product build, on the ISA these flags produce (a variant that fails it is not a measurement)
  # (MVHP_SKIP_ISA_CHECK=1: ablations that leave the stores out contradict the checker's store count by construction;
  #  such a build is a timing aid on crafted terms only and must never be installed)
  [ -n "$MVHP_SKIP_ISA_CHECK" ] && { echo "ISA check skipped for $b"; OBJS="$OBJS $o"; continue; }
  case $b in recon_quad|recon_oct) python3 $R/tools/check_prefetch_hazard.py $O/temps_$b/$b-hip-amdgcn-amd-amdhsa-gfx950.s > $O/$b.check || { echo "ISA CHECK FAILED for variant $NAME ($b)"; exit 1; } ;; esac
  OBJS="$OBJS $o"
done
HOST=$(ls $R/minivideo_amd/build/*.o | grep -v "\.hip\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 --hip-link -shared -fPIC -pthread -o $O/libminivideo.so $HOST $OBJS
[ -n "$MVHP_KEEP_TEMPS" ] || rm -rf $O/temps_* $O/*.o   # (65 MB per variant: the snapshot sent to the GPU box is capped)
echo built $O/libminivideo.so
