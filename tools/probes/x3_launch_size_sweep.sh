#!/bin/bash
run() { echo -n "$* -> "; env "$@" timeout -k 10 200 python bench.py --steps 1000 --warmup 100 --min-time 0.7 --node off --ddpg off --precision fp32 2>/dev/null | sed 's/.*"value": \([0-9.]*\).*/\1/'; }
run PDA_FC1_BLOCKS=128
run PDA_FC1_BLOCKS=256
run PDA_FC1_BLOCKS=512
run PDA_FC1_BLOCKS=1024
run PDA_PAIR_BLOCKS=256
run PDA_PAIR_BLOCKS=768
run PDA_PAIR_BLOCKS=1024
run PDA_X3_PAIR_MULT=2
echo -n "graph_steps=10 -> "; timeout -k 10 200 python bench.py --steps 1000 --warmup 100 --min-time 0.7 --node off --ddpg off --precision fp32 --graph-steps 10 2>/dev/null | sed 's/.*"value": \([0-9.]*\).*/\1/'
