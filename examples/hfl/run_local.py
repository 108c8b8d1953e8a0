"""Prepare the shards and start one FedAvg server and N clients on this
machine.

    python run_local.py --num_clients 3 -m 3 [-d non-iid] [--cpu]
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def main(argv=None):
    sys.path.insert(0, _HERE)
    from src import bank
    from src.proto.utils import parseargs
    args = parseargs(argv)
    data_dir = args.data_dir or os.path.join(_HERE, "data")
    os.makedirs(data_dir, exist_ok=True)
    source = os.path.join(data_dir, bank.SOURCE)
    if not os.path.isfile(source):
        print("no %s: generating a table of the same columns" % source)
        bank.synthesize(source)
    bank.split(args.num_clients, source, data_dir,
               non_iid=args.data_dist == "non-iid")
    env = dict(os.environ,
               PYTHONPATH=os.pathsep.join(
                   [_HERE, os.path.join(_HERE, "..", ".."),
                    os.environ.get("PYTHONPATH", "")]))
    common = ["--num_clients", str(args.num_clients), "--port",
              str(args.port), "-m", str(args.max_epoch)]
    procs = [subprocess.Popen([sys.executable, "-m", "src.server"] + common,
                              cwd=_HERE, env=env)]
    extra = ["--data-dir", data_dir, "-b", str(args.batch_size), "-l",
             str(args.lr), "-d", args.data_dist, "-p", args.precision,
             "--inner-epochs", str(args.inner_epochs)]
    if args.cpu:
        extra.append("--cpu")
    if not args.graph:
        extra.append("-g")
    for r in range(args.num_clients):
        procs.append(subprocess.Popen(
            [sys.executable, "-m", "src.client", "-i", str(r)] + common +
            extra, cwd=_HERE, env=env))
    codes = [p.wait(timeout=900) for p in procs]
    return max(codes)


if __name__ == "__main__":
    raise SystemExit(main())
