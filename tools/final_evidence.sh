#!/bin/bash
# The round's closing evidence on the GPU box (repo root), complete sets (bench line, traced bench + kernel stats, HBM traffic,
# SQ breakdown: tools/profile_full.sh) for every BASELINE configuration on the final build.
# usage: bash tools/final_evidence.sh <tag-prefix> [a|b]     a = configs 2 and 5 (2048, 64 and 512 pictures), b = configs 3 and 4
P=${1:-r04m}
PART=${2:-ab}
case $PART in *a*)
  bash tools/profile_full.sh ${P}_base1080 || exit 1
  bash tools/profile_full.sh ${P}_frames64 --frames 64 --steps 50 || exit 1
  bash tools/profile_full.sh ${P}_strong512 --strong 512 || exit 1 ;;
esac
case $PART in *b*)
  bash tools/profile_full.sh ${P}_high1080 --profile high || exit 1
  bash tools/profile_full.sh ${P}_high2160 --profile high --width-mbs 240 --height-mbs 135 --frames 1024 || exit 1 ;;
esac
