"""bench.py, process side: the watchdog around the parts of an N > 1 run that could hang, and the self-launch of N ranks when no
launcher set WORLD_SIZE."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Watchdog:
    """Host-side deadline around the parts of an N > 1 run that could hang (a candidate exchange of the start-up probe
    that never completes on some rank).  A hung collective cannot be cancelled from inside the process, so the run is built
    measure-first: the step is timed with the safe exchange BEFORE any other candidate is tried, and when a deadline
    passes every rank's own watchdog ends its process -- rank 0 after printing the record it already holds, with the
    reason in config.watchdog.  A hung candidate therefore costs that candidate, not the run."""

    def __init__(self, rank):
        import threading
        self.rank, self.what, self.deadline, self.fallback = rank, None, None, None
        self.record_printed = False  # the record is out: a deadline missed afterwards (teardown) must not fail the run
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def arm(self, what, seconds):
        with self._lock:
            self.what, self.deadline = what, time.monotonic() + seconds

    def disarm(self):
        with self._lock:
            self.what, self.deadline = None, None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                expired = self.deadline is not None and time.monotonic() > self.deadline
                what = self.what
            if expired:
                print(f"[rank {self.rank}] watchdog: '{what}' did not finish in time", file=sys.stderr, flush=True)
                if self.rank == 0 and self.fallback is not None:
                    rec = self.fallback(f"'{what}' did not finish within its deadline; reporting the measurement taken before it")
                    if rec is not None:
                        print(json.dumps(rec), flush=True)
                        os._exit(0)
                os._exit(0 if self.rank != 0 or self.record_printed else 3)


def self_launch(n):
    """`python bench.py --gpus N ...` started WITHOUT a launcher (N > 1, no WORLD_SIZE in the environment): start the N ranks
    ourselves -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
    bench.py <the same arguments>` -- and pass rank 0's JSON line (the children inherit stdout / stderr) and the job's exit
    code through.  A rank that fails makes torchrun stop the others and exit non-zero; its traceback is on stderr."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), *sys.argv[1:]]
    print("[bench] no launcher in the environment: " + " ".join(cmd), file=sys.stderr, flush=True)
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    return subprocess.call(cmd, env=env)


