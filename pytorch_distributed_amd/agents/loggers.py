"""Logger process: drains the shared accumulators on a timer and writes scalars
against `learner_step` (reference core/single_processes/dqn_logger.py:5-56 and
ddpg_logger.py, which differ only in the actor_loss line; one generic function
serves both here).

Tags kept from the reference: evaluator/{avg_steps,avg_reward,repisodes_solved,
nepisodes}, actor/{avg_steps,avg_reward,repisodes_solved,total_nframes},
learner/critic_loss (+ learner/actor_loss for DDPG).  Added: throughput/
{learner_updates_per_sec, actor_frames_per_sec} derived from the global step
counters -- the metric BASELINE.json names, which the reference never computes.
`actor/total_nframes` uses a fresh counter read (the reference plots a value
refreshed only when an evaluator report arrives, dqn_logger.py:24,42).
"""
import time

from ..utils.board import Board


def _logger_loop(process_ind, args, global_logs, actor_logs, learner_logs, evaluator_logs):
    print("---------------------------->", process_ind, "logger")
    ap = args.agent_params
    board = Board(args.log_dir)
    poll = float(getattr(ap, "poll_interval", 5.0))
    last_log_time = time.time()
    last_learner_step, last_actor_step = 0, 0

    def drain(final=False):
        nonlocal last_log_time, last_learner_step, last_actor_step
        if evaluator_logs.logger_lock.value:
            with evaluator_logs.logger_lock.get_lock():
                n = evaluator_logs.nepisodes.value
                if n > 0:
                    ls = global_logs.learner_step.value
                    board.add_scalar("evaluator/avg_steps", evaluator_logs.total_steps.value / n, ls)
                    board.add_scalar("evaluator/avg_reward", evaluator_logs.total_reward.value / n, ls)
                    board.add_scalar("evaluator/repisodes_solved", evaluator_logs.nepisodes_solved.value / n, ls)
                    board.add_scalar("evaluator/nepisodes", n, ls)
                evaluator_logs.logger_lock.value = False
        now = time.time()
        if final or now - last_log_time > ap.logger_freq:
            ls = global_logs.learner_step.value
            acs = global_logs.actor_step.value
            with actor_logs.nepisodes.get_lock():
                n = actor_logs.nepisodes.value
                if n > 0:
                    board.add_scalar("actor/avg_steps", actor_logs.total_steps.value / n, ls)
                    board.add_scalar("actor/avg_reward", actor_logs.total_reward.value / n, ls)
                    board.add_scalar("actor/repisodes_solved", actor_logs.nepisodes_solved.value / n, ls)
                    board.add_scalar("actor/total_nframes", acs, ls)
                    actor_logs.total_steps.value = 0
                    actor_logs.total_reward.value = 0.0
                    actor_logs.nepisodes.value = 0
                    actor_logs.nepisodes_solved.value = 0
            with learner_logs.loss_counter.get_lock():
                c = learner_logs.loss_counter.value
                if c > 0:
                    if hasattr(learner_logs, "actor_loss"):
                        board.add_scalar("learner/actor_loss", learner_logs.actor_loss.value / c, ls)
                        learner_logs.actor_loss.value = 0.0
                    board.add_scalar("learner/critic_loss", learner_logs.critic_loss.value / c, ls)
                    learner_logs.critic_loss.value = 0.0
                    learner_logs.loss_counter.value = 0
            dt = max(now - last_log_time, 1e-6)
            board.add_scalar("throughput/learner_updates_per_sec", (ls - last_learner_step) / dt, ls)
            board.add_scalar("throughput/actor_frames_per_sec", (acs - last_actor_step) / dt, ls)
            last_learner_step, last_actor_step, last_log_time = ls, acs, now

    while global_logs.running(ap.steps):
        time.sleep(poll)
        drain()
    drain(final=True)
    board.close()


def dqn_logger(process_ind, args, global_loggers, actor_loggers, learner_loggers, evaluator_loggers):
    _logger_loop(process_ind, args, global_loggers, actor_loggers, learner_loggers, evaluator_loggers)


def ddpg_logger(process_ind, args, global_loggers, actor_loggers, learner_loggers, evaluator_loggers):
    _logger_loop(process_ind, args, global_loggers, actor_loggers, learner_loggers, evaluator_loggers)
