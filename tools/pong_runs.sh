#!/bin/bash
# usage: pong_runs.sh <outdir> <core> <seed...>   -- one bench/pong_wallclock.py run per seed, each under its own timeout
out=$1; core=$2; shift 2
mkdir -p $out
for seed in "$@"; do
  timeout -k 10 ${PONG_TIMEOUT:-240} python bench/pong_wallclock.py --core $core --seed $seed ${PONG_ARGS} > $out/${core}_s${seed}.txt 2> $out/${core}_s${seed}.err
  echo "== $core seed $seed: rc $?"; tail -1 $out/${core}_s${seed}.txt | cut -c 1-420
done
