"""Parts of bench.py (the entry point stays /bench.py: the driver runs it by that name): counters (record side), context
(workload side), launch (watchdog, self-launch)."""
