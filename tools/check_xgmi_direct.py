"""The direct xGMI collectives (csrc/hip/xgmi.hip, parallel/xgmi.py) with
several ranks on ONE GPU: every rank is a process of its own, maps the other
ranks' windows through IPC handles and runs the same kernels a multi-GPU node
would run (the peers' memory is then local instead of behind a link).  Checks

* all-reduce, one-shot and two-shot, fp32 / fp16 / bf16, sizes around the
  vector, slice and one-shot boundaries, against a float64 sum of the ranks'
  inputs computed on the host; every rank must hold bit-identical results;
* all-gather with 16-byte and 4-byte granularity;
* the same collectives captured into a hipGraph and replayed;
* that a rank whose peer never arrives leaves its kernel after the time limit
  with the error word set (``XG_CASE=timeout``).

    python tools/check_xgmi_direct.py             # launches the ranks itself
"""
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NRANKS = int(os.environ.get("XG_RANKS", "4"))
CASE = os.environ.get("XG_CASE", "collectives")
sys.path.insert(0, ROOT)


def _input(rank, n, seed):
    rs = np.random.RandomState(1000 * seed + rank)
    # multiples of 1/64 in [-4, 4): sums of 8 of them are exact in fp16
    return rs.randint(-256, 256, n).astype(np.float64) / 64.0


def rank_main():
    import torch
    import torch.distributed as dist
    from singa_amd.parallel import xgmi
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group(
        "gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"],
        rank=rank, world_size=world)

    def exchange(obj):
        box = [None] * world
        dist.all_gather_object(box, obj)
        return box

    dev = torch.device("cuda", 0)
    limit = 3.0 if CASE == "timeout" else 30.0
    win = xgmi.XgmiWindow(rank, world, dev, 96 << 20, exchange, timeout=limit)
    if CASE == "timeout":
        t = win.tensor(1024, torch.float32)
        t.fill_(1.0)
        torch.cuda.synchronize()
        dist.barrier()
        if rank != 0:                  # rank 0 never joins the collective
            win.all_reduce(t)
            torch.cuda.synchronize()   # returns: the wait inside is bounded
            try:
                win.check()
                print("rank %d: no error raised" % rank, flush=True)
                return 1
            except RuntimeError as e:
                assert "rank 0" in str(e), str(e)
        dist.barrier()
        sys.stdout.write("rank %d XGMI-RANK-OK 0\n" % rank)
        sys.stdout.flush()
        win._err_host and win.k.xgmi_clear_word(win._err_host)
        win.close()
        dist.destroy_process_group()
        return 0

    digest = 0
    sizes = [4, 5, 1000, 4096, 65536, 65537, 262144 + 12, 1 << 20,
             5 * (1 << 20) + 3]
    seed = 0
    for tdt, tol in ((torch.float32, 0.0), (torch.float16, 0.0),
                     (torch.bfloat16, 0.05)):
        for n in sizes:
            seed += 1
            t = win.tensor(n, tdt)
            want = sum(_input(r, n, seed) for r in range(world))
            mine = torch.from_numpy(_input(rank, n, seed)).to(tdt).to(dev)
            for rep in range(3):       # the barrier epochs advance
                t.copy_(mine)
                win.all_reduce(t)
                torch.cuda.synchronize()
                got = t.double().cpu().numpy()
                err = np.abs(got - want).max()
                assert err <= tol * max(1.0, np.abs(want).max()), (
                    str(tdt), n, rep, err)
            digest = zlib.crc32(t.view(torch.uint8).cpu().numpy().tobytes()
                                if tdt != torch.bfloat16 else
                                t.view(torch.int16).cpu().numpy().tobytes(),
                                digest)
    # all-gather: 16-byte vectors and single words
    for words in (4, 1 + 2 * 37, 4096, 100001):
        seed += 1
        src = win.tensor(words, torch.int32)
        out = win.tensor(world * words, torch.int32)
        src.copy_(torch.arange(words, dtype=torch.int32) * (rank + 1) + seed)
        win.all_gather(out, src)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(world, words)
        for r in range(world):
            assert np.array_equal(
                got[r], np.arange(words, dtype=np.int32) * (r + 1) + seed), (
                    words, r)
    # all-to-all: chunk p of the result came from rank p, which held it as its
    # chunk number `rank`
    for words in (4, 3, 1024, 50001):
        seed += 1
        src = win.tensor(world * words, torch.int32)
        out = win.tensor(world * words, torch.int32)
        base = torch.arange(world * words, dtype=torch.int32)
        for rep in range(2):
            src.copy_(base + 1000003 * rank + seed + rep)
            win.all_to_all(out, src)
            torch.cuda.synchronize()
            got = out.cpu().numpy().reshape(world, words)
            for r in range(world):
                want = np.arange(rank * words, (rank + 1) * words,
                                 dtype=np.int32) + 1000003 * r + seed + rep
                assert np.array_equal(got[r], want), (words, r, rep)
    # reduce-scatter: this rank's slice holds the sum, the others the input
    for tdt, tol in ((torch.float32, 0.0), (torch.float16, 0.0)):
        for per in (8, 4096, 262144 + 8):
            seed += 1
            n = per * world
            t = win.tensor(n, tdt)
            want = sum(_input(r, n, seed) for r in range(world))
            mine = torch.from_numpy(_input(rank, n, seed)).to(tdt).to(dev)
            for rep in range(2):
                t.copy_(mine)
                part = win.reduce_scatter(t)
                torch.cuda.synchronize()
                got = part.double().cpu().numpy()
                ref = want[rank * per:(rank + 1) * per]
                assert np.abs(got - ref).max() <= tol, (str(tdt), per, rep)
                other = (rank + 1) % world
                if other != rank:
                    assert torch.equal(t[other * per:(other + 1) * per],
                                       mine[other * per:(other + 1) * per])
    # captured into a hipGraph and replayed
    big = win.tensor(3 * (1 << 20) + 8, torch.float32)
    small = win.tensor(2048, torch.float32)
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream(device=dev)
    side = torch.cuda.Stream(device=dev)
    big.fill_(1.0)
    small.fill_(1.0)
    torch.cuda.synchronize()
    dist.barrier()
    with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        big.mul_(1.0)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            win.all_reduce(big)
            win.all_reduce(small)
        torch.cuda.current_stream(dev).wait_stream(side)
        big.mul_(0.5)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    win.check()
    # x -> x * W / 2 per replay for big, x * W for small
    want_big = (world / 2.0) ** 3
    want_small = float(world) ** 3
    assert abs(float(big[0]) - want_big) <= 1e-6 * want_big and \
        abs(float(big[-1]) - want_big) <= 1e-6 * want_big, (float(big[0]),
                                                            want_big)
    assert abs(float(small[-1]) - want_small) <= 1e-6 * want_small
    win.check()
    dist.barrier()
    sys.stdout.write("rank %d XGMI-RANK-OK %d\n" % (rank, digest))
    sys.stdout.flush()
    win.close()
    dist.destroy_process_group()
    return 0


def main():
    if "RANK" in os.environ:
        return rank_main()
    log = os.path.join(ROOT, "gpurun_out", "xgmi_ranks_%s.log" % CASE)
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as fd:
        rc = subprocess.run(
            [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1",
             "--nproc-per-node=%d" % NRANKS, "--master-addr", "127.0.0.1",
             "--master-port", "29547", os.path.abspath(__file__)],
            env=dict(os.environ), stdout=fd, stderr=subprocess.STDOUT,
            timeout=240).returncode
    text = open(log).read()
    print(text[-3000:])
    if rc != 0:
        return 1
    import re
    digests = re.findall(r"rank \d+ XGMI-RANK-OK (\d+)", text)
    assert len(digests) == NRANKS, digests
    assert len(set(digests)) == 1, "ranks hold different bits: %s" % digests
    print("XGMI-DIRECT-OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
