"""All-reduce bandwidth / latency over message sizes on the ranks of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
        --master-addr 127.0.0.1 --master-port 29611 tools/bench_allreduce.py
    SINGA_DIST_FORCE=1 python tools/bench_allreduce.py      # one rank: the
                                                            # launch floor

For every size the ways of summing a float32 buffer over the ranks are timed:
the RCCL call on the caller's stream that the communicator uses,
reduce-scatter + all-gather on the same communicator, the RCCL call as a
captured hipGraph replay (the way a training step issues it), the direct
kernel that reads every peer's window over its own link (SINGA_XGMI_DIRECT=1,
csrc/hip/xgmi.hip; several ranks may share one GPU with SINGA_DIST_BACKEND=gloo)
and -- for reference -- the host-staged exchange over the gloo control group
that serves when RCCL is switched off.
Reported: microseconds per call, algorithm bandwidth (bytes / time) and bus
bandwidth (algorithm bandwidth x 2 (N - 1) / N, the per-link load of a ring).
xGMI gives every GPU 7 links of about 153 GB/s peer to peer."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from singa_amd import device  # noqa: E402
from singa_amd.parallel.communicator import Communicator  # noqa: E402


def timed(fn, iters, dev):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-log2", type=int, default=12)
    ap.add_argument("--max-log2", type=int, default=27)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # a rehearsal with several ranks per GPU wraps around the devices
    local %= max(torch.cuda.device_count(), 1)
    dev = device.create_cuda_gpu_on(local)
    torch.cuda.set_device(local)
    comm = Communicator(1 << 22)       # rendezvous from the environment
    td = dev.torch_device
    world, rank = comm.world_size, comm.global_rank
    rccl = comm._rccl if getattr(comm, "_rccl", None) is not None else None
    rows = []
    for lg in range(args.min_log2, args.max_log2 + 1):
        n = (1 << lg) // 4
        n -= n % max(world, 1)
        buf = torch.ones(n, dtype=torch.float32, device=td)
        res = {"bytes": n * 4}

        win = getattr(comm, "_xgmi", None)
        if win is not None and n * 4 + 4096 <= win.bytes // 2:
            if "cursor0" not in globals():
                globals()["cursor0"] = win._cursor
            win._cursor = globals()["cursor0"]      # one buffer, reused
            wbuf = win.tensor(n, torch.float32)
            wbuf.fill_(1.0)
            res["xgmi_us"] = timed(lambda: win.all_reduce(wbuf), args.iters,
                                   td) * 1e6
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=td)
            torch.cuda.synchronize(td)
            if dist.is_initialized():
                dist.barrier()
            with torch.cuda.graph(g, stream=side,
                                  capture_error_mode="thread_local"):
                for _ in range(8):
                    win.all_reduce(wbuf)
            res["xgmi_graph_us"] = timed(g.replay, max(args.iters // 4, 2),
                                         td) * 1e6 / 8
            win.check()
        if (rccl is None and win is None) or lg <= 20:
            def staged_call():
                host = buf.cpu()
                if dist.is_initialized():
                    dist.all_reduce(host)
                buf.copy_(host)
            res["host_staged_us"] = timed(staged_call, 3, td) * 1e6
        if rccl is not None:
            res["direct_us"] = timed(lambda: rccl.all_reduce(buf),
                                     args.iters, td) * 1e6
            res["two_phase_us"] = timed(
                lambda: rccl.all_reduce_two_phase(buf, world, rank),
                args.iters, td) * 1e6
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=td)
            side.wait_stream(torch.cuda.current_stream(td))
            with torch.cuda.stream(side):
                rccl.all_reduce(buf)
                torch.cuda.synchronize(td)
                with torch.cuda.graph(g, stream=side):
                    for _ in range(8):
                        rccl.all_reduce(buf)
            res["graph_us"] = timed(g.replay, max(args.iters // 4, 2),
                                    td) * 1e6 / 8
        best = min(v for k, v in res.items() if k.endswith("_us"))
        res["algbw_GBps"] = res["bytes"] / best / 1e3
        res["busbw_GBps"] = res["algbw_GBps"] * 2 * (world - 1) / max(world, 1)
        rows.append(res)
        if rank == 0:
            print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v)
                              for k, v in res.items()}), flush=True)
    if rank == 0:
        print(json.dumps({"world": world, "peak_busbw_GBps": round(max(
            r["busbw_GBps"] for r in rows), 1)}))


if __name__ == "__main__":
    main()
