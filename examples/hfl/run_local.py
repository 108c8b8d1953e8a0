"""Start one FedAvg server and N clients on this machine.

    python run_local.py --num_clients 2 -m 3
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv=None):
    sys.path.insert(0, _HERE)
    from src.server import parseargs
    args = parseargs(argv)
    env = dict(os.environ,
               PYTHONPATH=os.pathsep.join(
                   [_HERE, os.path.join(_HERE, "..", ".."),
                    os.environ.get("PYTHONPATH", "")]))
    common = ["--num_clients", str(args.num_clients), "--port",
              str(args.port), "-m", str(args.max_epoch)]
    procs = [subprocess.Popen([sys.executable, "-m", "src.server"] + common,
                              cwd=_HERE, env=env)]
    for r in range(args.num_clients):
        procs.append(subprocess.Popen(
            [sys.executable, "-m", "src.client", "-i", str(r)] + common,
            cwd=_HERE, env=env))
    codes = [p.wait(timeout=600) for p in procs]
    return max(codes)


if __name__ == "__main__":
    raise SystemExit(main())
