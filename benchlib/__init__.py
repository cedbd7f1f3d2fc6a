"""bench.py's parts.  timed.py is the timed path (build, step, measure, the JSON line); everything else is around it:
cli (flags), guards (watchdog, budget, line in hand), transports (device-to-device transports and their trials), search (route
candidates of a multi-GPU run), checks (closed-form result checks), secondary (the other workloads of the reference's harness on
one GPU), baseline (the reference on the host cores, live counter traffic), launcher (python bench.py --gpus N as typed), run
(the order of it all)."""
