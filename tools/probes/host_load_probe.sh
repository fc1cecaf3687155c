#!/bin/bash
# Does host CPU load alone slow the graph-replayed learner?  N busy-loop processes (nice 5, like the actors) beside the
# standalone benchmark.  usage: host_load_probe.sh <N burners>
N=${1:-12}
pids=""
for i in $(seq 1 $N); do
  nice -n 5 timeout 60 python3 -c "while True: pass" &
  pids="$pids $!"
done
sleep 1
python3 bench.py --node off --precision fp32 --min-time 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('burners', $N, 'updates/s', d['value'], 'ms/step min/max', d['ms_per_step_min'], d['ms_per_step_max'])"
for p in $pids; do kill $p 2>/dev/null; done
wait 2>/dev/null
