"""Start a data-parallel job, one process per rank, and start it again when
it fails.

    python tools/launch_restart.py --nproc 8 [--max-restarts 3]
        [--master-port 29500] train.py --resume ...

Every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
and, unless the caller set them, the failure-detection switches of
``singa_amd.parallel.health`` (heartbeat every second, abort with status 75
when a peer is lost).  When any rank ends with a non-zero status the others
are stopped (the processes this launcher started, by pid), and the whole job
is started again -- on a fresh rendezvous port, with SINGA_RESTART_COUNT set
and without SINGA_FAULT_INJECT (an injected fault is for the first attempt) --
until it succeeds or ``--max-restarts`` is used up.  The training script is
expected to resume from its newest checkpoint
(``singa_amd.utils.checkpoint.CheckpointManager``).

The launcher itself never touches a GPU.
"""
import argparse
import os
import subprocess
import sys
import time


def run_once(args, generation, env_base):
    port = args.master_port + 2 * generation
    procs = []
    for rank in range(args.nproc):
        env = dict(env_base, RANK=str(rank), LOCAL_RANK=str(rank),
                   WORLD_SIZE=str(args.nproc), MASTER_ADDR=args.master_addr,
                   MASTER_PORT=str(port), SINGA_RESTART_COUNT=str(generation),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("SINGA_HEARTBEAT", "1")
        env.setdefault("SINGA_ON_RANK_FAILURE", "abort")
        if generation > 0:
            env.pop("SINGA_FAULT_INJECT", None)
        procs.append(subprocess.Popen([sys.executable] + args.command,
                                      env=env))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for rank, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0):
                failed = (rank, rc)
                break
        time.sleep(0.1)
    if failed is None:
        for rank, p in enumerate(procs):
            if p.returncode != 0:
                failed = (rank, p.returncode)
                break
    if failed is not None:
        # give the survivors a moment to notice by themselves (status 75),
        # then stop what is left
        deadline = time.time() + args.grace
        while time.time() < deadline and any(p.poll() is None for p in procs):
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return failed, [p.returncode for p in procs]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--nproc", type=int, required=True)
    ap.add_argument("--max-restarts", type=int, default=3)
    ap.add_argument("--master-addr", default="127.0.0.1")
    ap.add_argument("--master-port", type=int, default=29500)
    ap.add_argument("--grace", type=float, default=15.0,
                    help="seconds the other ranks get to end by themselves "
                         "after one failed")
    ap.add_argument("command", nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    if args.command and args.command[0] == "--":
        args.command = args.command[1:]
    if not args.command:
        ap.error("no training command given")
    env_base = dict(os.environ)
    for generation in range(args.max_restarts + 1):
        failed, codes = run_once(args, generation, env_base)
        if failed is None:
            print("[launcher] job finished (attempt %d)" % (generation + 1),
                  flush=True)
            return 0
        print("[launcher] attempt %d: rank %d ended with status %d (all: %s)"
              % (generation + 1, failed[0], failed[1], codes), flush=True)
    print("[launcher] giving up after %d attempts" % (args.max_restarts + 1),
          flush=True)
    return 1


if __name__ == "__main__":
    raise SystemExit(main())
