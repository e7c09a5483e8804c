#!/bin/bash
# blocking / pipelined Groth16 and MSM against the placement of the library's queues: bash tools/dbg/pad_sweep.sh out_dir [placement]
#   KG_STREAM_PAD puts never-used streams in front of the context's queues; KG_QUEUE_PLACEMENT=0 switches the placement probe off
O=${1:-gpurun_out/pad}; mkdir -p $O
for pl in ${2:-1 0}; do
for pad in 0 1 2 3; do
  KG_QUEUE_PLACEMENT=$pl KG_STREAM_PAD=0,$pad python bench.py --no-cpu-baseline --no-nova --no-ntt > $O/b_pl${pl}_pad$pad.json 2>$O/err_pl${pl}_pad$pad.txt
done; done
