#!/bin/bash
# The pixel game `catch` in the full job under different replay ratios / env counts: is the evaluator's oscillation
# (profiles/r2_catch_precision_curves.txt) a replay-ratio effect?   usage: catch_ratio_sweep.sh <outdir>
out=$1; mkdir -p $out
run() { tag=$1; shift
  timeout -k 10 150 python bench/learning_sanity.py --config 2 --steps 150000 --set game=catch env_params.game=catch \
      agent_params.learn_start=1000 agent_params.evaluator_nepisodes=20 agent_params.evaluator_freq=1.0 precision=fp32 "$@" \
      > $out/catch_$tag.txt 2>&1
  echo "== $tag rc=$?"; grep -i "avg reward\|eval" $out/catch_$tag.txt | tail -3 | cut -c 1-300; tail -1 $out/catch_$tag.txt | cut -c 1-400; }
run r8_e1 agent_params.max_replay_ratio=0.125
run r16_e4 agent_params.max_replay_ratio=0.0625 num_envs_per_actor=4
run r64_e4 agent_params.max_replay_ratio=0.015625 num_envs_per_actor=4
