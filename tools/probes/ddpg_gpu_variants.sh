mkdir -p gpurun_out/ddpg
run() { tag=$1; shift; for s in 100 101 102 103; do timeout -k 10 200 python bench/learning_sanity.py --config 7 --steps 100000 --set seed=$s agent_params.evaluator_freq=1.0 "$@" > gpurun_out/ddpg/${tag}_$s.log 2>&1 || exit 1; echo $tag seed $s: $(grep "eval @" gpurun_out/ddpg/${tag}_$s.log | awk '{print $NF}' | tr "\n" " "); done; }
run ls5k agent_params.learn_start=5000
run sync5 agent_params.actor_sync_freq=5 param_sync_freq=5
run r025 agent_params.max_replay_ratio=0.25 agent_params.min_replay_ratio=0.25
