"""GPU probe: can this stack capture (a) an RCCL all-reduce issued on a side
stream and (b) timing events into one hipGraph, and replay them?

Prints one line per capability; run it on one MI355X:
    python tools/probe_capture.py
"""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

import torch
import torch.distributed as dist


def probe_rccl():
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    x = torch.ones(1 << 20, device=dev)
    side = torch.cuda.Stream(device=dev)
    dist.all_reduce(x)            # communicator set-up happens eagerly
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        x.mul_(2.0)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dist.all_reduce(x)
        torch.cuda.current_stream().wait_stream(side)
        x.add_(1.0)
    x.fill_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    want = 1.0
    for _ in range(3):
        want = want * 2.0 + 1.0
    ok = bool((x == want).all())
    print("rccl_capture:", "ok" if ok else "wrong %r" % x[:4].tolist())
    dist.destroy_process_group()


def probe_events():
    dev = torch.device("cuda", 0)
    x = torch.ones(1 << 24, device=dev)
    try:
        e0 = torch.cuda.Event(enable_timing=True, external=True)
        e1 = torch.cuda.Event(enable_timing=True, external=True)
    except TypeError as e:
        print("external_events: unsupported (%s)" % e)
        return
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        e0.record()
        for _ in range(20):
            x.mul_(1.0001)
        e1.record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("external_events: ok elapsed %.3f ms" % e0.elapsed_time(e1))


if __name__ == "__main__":
    for fn in (probe_rccl, probe_events):
        try:
            fn()
        except Exception as e:  # report and go on with the next probe
            print("%s: FAILED %s: %s" % (fn.__name__, type(e).__name__, e))
            sys.stdout.flush()
