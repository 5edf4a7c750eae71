#!/bin/bash
# The round's closing evidence on the GPU box (repo root): the complete set for config 2 on the final bench.py, and fresh
# bench lines (end-to-end legs included) for configs 3, 4 and 5.  usage: bash tools/final_evidence.sh <tag-prefix>
P=${1:-r04h}
bash tools/profile_full.sh ${P}_base1080 || exit 1
bash tools/profile_full.sh ${P}_frames64 --frames 64 --steps 50 || exit 1
timeout -k 10 500 python3 bench.py --profile high > gpurun_out/${P}_high1080_bench.json 2> gpurun_out/${P}_high1080_bench.err || exit 1
echo "high1080 done"
timeout -k 10 600 python3 bench.py --profile high --width-mbs 240 --height-mbs 135 --frames 1024 > gpurun_out/${P}_high2160_bench.json 2> gpurun_out/${P}_high2160_bench.err || exit 1
echo "high2160 done"
bash tools/profile_full.sh ${P}_strong512 --strong 512 || exit 1
