"""Golden tests of the native graph scheduler (singa_amd/csrc/runtime/graph.cc).

The expected node / edge / block counts, block types, reference counts,
execution plan and per-node recycling lists are those of the reference's
test/singa/test_scheduler.cc.
"""
import gc

import numpy as np
import pytest

from singa_amd import _rt
from singa_amd.core import tensor_core as tc
from singa_amd.core.device import get_default_device

BT = _rt.BlockType


def nop():
    pass


def blk():
    return _rt.Block(4)


def info(g, b):
    return g.block_info(b)


def test_add_op():
    g = _rt.Graph()
    a, o = blk(), blk()
    g.AddOperation(nop, [a], [o], "op")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (1, 2, 2)
    assert g.leaf_block_ids() == [o.id()]
    assert g.node_in_edges(0) == [(-1, 0, a.id())]
    assert g.node_out_edges(0) == [(0, -1, o.id())]
    assert g.edge(0) == (-1, 0, a.id())
    assert g.edge(1) == (0, -1, o.id())
    ia, io = info(g, a), info(g, o)
    assert (ia["id"], ia["type"], ia["graph_ref"], ia["write_edge"]) == \
        (0, BT.kInput, 1, None)
    assert (io["id"], io["type"], io["graph_ref"], io["write_edge"]) == \
        (1, BT.kEnd, 1, (0, -1))
    assert g.dirty()


def test_add_sync_op():
    g = _rt.Graph()
    a, o = blk(), blk()
    g.AddOperation(nop, [a], [o], "op")
    g.AddOperation(nop, [], [], "sync")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (2, 3, 2)
    assert g.leaf_block_ids() == [o.id()]
    assert g.node_out_edges(0) == [(0, 1, o.id())]
    assert g.node_in_edges(1) == [(0, 1, o.id())]
    assert g.node_out_edges(1) == [(1, -1, o.id())]
    io = info(g, o)
    assert (io["type"], io["graph_ref"], io["write_edge"]) == \
        (BT.kInter, 1, (1, -1))


def test_add_inplace_op():
    g = _rt.Graph()
    a, o = blk(), blk()
    g.AddOperation(nop, [a], [a], "op")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (1, 2, 1)
    ia = info(g, a)
    assert (ia["type"], ia["graph_ref"], ia["write_edge"]) == \
        (BT.kParam, 2, (0, -1))
    assert g.leaf_block_ids() == [a.id()]
    g.AddOperation(nop, [a, o], [o], "op")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (2, 4, 2)
    assert g.node_in_edges(1) == [(0, 1, a.id()), (-1, 1, o.id())]
    assert g.node_out_edges(1) == [(1, -1, o.id())]
    ia, io = info(g, a), info(g, o)
    assert (ia["type"], ia["graph_ref"], ia["write_edge"]) == \
        (BT.kParam, 3, (0, 1))
    assert (io["type"], io["graph_ref"], io["write_edge"]) == \
        (BT.kParam, 2, (1, -1))
    assert g.leaf_block_ids() == [o.id()]


def test_block_type_param():
    g = _rt.Graph()
    a, mid, o = blk(), blk(), blk()
    g.AddOperation(nop, [a], [a], "op")
    g.AddOperation(nop, [a, mid], [o], "op")
    g.AddOperation(nop, [o], [mid], "op")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (3, 5, 3)
    assert len(g.leaf_block_ids()) == 1
    ia, im = info(g, a), info(g, mid)
    assert (ia["id"], ia["type"], ia["graph_ref"]) == (0, BT.kParam, 3)
    assert (im["id"], im["type"], im["graph_ref"]) == (1, BT.kParam, 2)
    assert im["write_edge"] == (2, -1)


def test_block_type_inter():
    g = _rt.Graph()
    a, mid, o = blk(), blk(), blk()
    g.AddOperation(nop, [a], [mid, o], "op")
    g.AddOperation(nop, [mid], [], "op")
    g.AddOperation(nop, [o], [o], "op")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (3, 4, 3)
    im, io = info(g, mid), info(g, o)
    assert (im["id"], im["type"], im["graph_ref"]) == (1, BT.kInter, 2)
    assert (io["id"], io["type"], io["graph_ref"]) == (2, BT.kInter, 3)


def test_block_type_end():
    g = _rt.Graph()
    a, o1, o2 = blk(), blk(), blk()
    g.AddOperation(nop, [a], [o1], "op")
    g.AddOperation(nop, [], [o2], "op")
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (2, 3, 3)
    assert len(g.leaf_block_ids()) == 2
    assert info(g, o1)["type"] == BT.kEnd and info(g, o2)["type"] == BT.kEnd
    assert info(g, o1)["graph_ref"] == 1 and info(g, o2)["graph_ref"] == 1


def _fwd_bwd_graph(dev, g):
    """(in + b1) * in + b2 and its hand-written backward."""
    T = lambda: tc.Tensor([1], dev)
    names = "x mid out b1 b2 dx dx1 dx2 dy1 dy2 db1 db2".split()
    t = {n: T() for n in names}
    B = lambda *ns: [t[n].block() for n in ns]

    # every op owns a shallow copy of each operand, like the reference's
    # lambdas that capture tensors by value: a block whose only owners are
    # graph ops may be recycled, one the caller still holds may not
    def own(*ns):
        return [tc.Tensor(t[n]) for n in ns]

    def cp(dst, src):
        dst.flat().copy_(src.flat())

    def binop(kind, a, b, o):
        x, y, z = own(a, b, o)
        return lambda: dev.math.binary(kind, x.view(), y.view(), z.view())

    def copy2(src, d1, d2):
        s_, a_, b_ = own(src, d1, d2)

        def f():
            cp(a_, s_)
            cp(b_, s_)
        return f

    op1 = binop("add", "x", "b1", "mid")
    op2 = binop("mul", "mid", "x", "out")
    op3 = binop("add", "out", "b2", "out")
    op4 = copy2("out", "dy1", "db2")
    op5a = binop("mul", "dy1", "x", "dy2")
    op5b = binop("mul", "dy1", "mid", "dx1")

    def op5():
        op5a()
        op5b()

    op6 = copy2("dy2", "dx2", "db1")
    op7 = binop("add", "dx1", "dx2", "dx")

    g.AddOperation(op1, B("x", "b1"), B("mid"), "op1")
    g.AddOperation(op2, B("mid", "x"), B("out"), "op2")
    g.AddOperation(op3, B("out", "b2"), B("out"), "op3")
    g.AddOperation(op4, B("out"), B("dy1", "db2"), "op4")
    g.AddOperation(op5, B("dy1"), B("dy2", "dx1"), "op5")
    g.AddOperation(op6, B("dy2"), B("dx2", "db1"), "op6")
    g.AddOperation(op7, B("dx1", "dx2"), B("dx"), "op7")
    t["x"].SetValue(0)
    t["b1"].SetValue(-1)
    t["b2"].SetValue(2)
    return t


@pytest.mark.parametrize("serial", [False, True])
def test_run_graph(serial):
    dev = get_default_device()
    g = _rt.Graph()
    t = _fwd_bwd_graph(dev, g)
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (7, 14, 12)
    assert len(g.leaf_block_ids()) == 3
    if serial:
        g.RunInSerial()
    else:
        g.RunGraph()
    assert t["dx"].GetFloatValue(1)[0] == -2
    assert t["db1"].GetFloatValue(1)[0] == 0
    assert t["db2"].GetFloatValue(1)[0] == 2


def test_multiple_independent_ops():
    g = _rt.Graph()
    ws, b1, b2, b3, b4 = blk(), blk(), blk(), blk(), blk()
    for b in (b1, b2, b3, b4):
        g.AddOperation(nop, [], [ws], "clean")
        g.AddOperation(nop, [b], [ws, b], "use")
    assert g.num_nodes() == 8
    g.RunGraph()


def test_auto_recycle():
    """Blocks referenced only by the graph are released right after the last
    node that uses them (reference test_scheduler.cc:633-760)."""
    dev = get_default_device()
    g = _rt.Graph()
    order = ("x b1 mid1 out b2 mid2 dy1 dy2 dy3 dx1 dx2 db1 dx3 db2 dx").split()
    keep = {}

    def build():
        t = {n: tc.Tensor([1], dev) for n in order}
        # the closures own shallow copies, as recorded device ops do
        c = {n: tc.Tensor(v) for n, v in t.items()}
        m = dev.math

        def add(a, b, o):
            return lambda: m.binary("add", c[a].view(), c[b].view(),
                                    c[o].view())

        def mul(a, b, o):
            return lambda: m.binary("mul", c[a].view(), c[b].view(),
                                    c[o].view())

        def cp2(src, d1, d2):
            def f():
                c[d1].flat().copy_(c[src].flat())
                c[d2].flat().copy_(c[src].flat())
            return f

        def two(f1, f2):
            def f():
                f1()
                f2()
            return f

        B = lambda *ns: [t[n].block() for n in ns]
        g.AddOperation(add("x", "b1", "mid1"), B("x", "b1"), B("mid1"), "op1")
        g.AddOperation(mul("mid1", "x", "out"), B("mid1", "x"), B("out"), "op2")
        g.AddOperation(add("x", "b2", "mid2"), B("x", "b2"), B("mid2"), "op3")
        g.AddOperation(add("out", "mid2", "out"), B("out", "mid2"), B("out"),
                       "op4")
        g.AddOperation(cp2("out", "dy1", "dy2"), B("out"), B("dy1", "dy2"),
                       "op5")
        g.AddOperation(two(mul("dy1", "x", "dy3"), mul("dy1", "mid1", "dx1")),
                       B("x", "mid1", "dy1"), B("dy3", "dx1"), "op6")
        g.AddOperation(cp2("dy3", "dx2", "db1"), B("dy3"), B("dx2", "db1"),
                       "op7")
        g.AddOperation(cp2("dy2", "dx3", "db2"), B("dy2"), B("dx3", "db2"),
                       "op8")
        g.AddOperation(add("dx1", "dx2", "dx"), B("dx1", "dx2"), B("dx"), "op9")
        g.AddOperation(add("dx", "dx3", "dx"), B("dx", "dx3"), B("dx"), "op10")
        t["x"].SetValue(0)
        t["b1"].SetValue(-1)
        t["b2"].SetValue(2)
        blocks = {n: t[n].block() for n in order}
        # touch every block so that "initialized" reflects recycling only
        for n in order:
            t[n].flat()
        # in, b1, b2 stay alive outside the graph, like the model's inputs
        # and parameters; everything else is owned by the graph alone
        keep.update({n: t[n] for n in ("x", "b1", "b2")})
        return blocks

    blocks = build()
    gc.collect()
    assert (g.num_nodes(), g.num_edges(), g.num_blocks()) == (10, 21, 15)
    assert len(g.leaf_block_ids()) == 3
    g.RunGraph()
    assert not g.dirty()
    assert g.begin_nodes() == [0, 2]
    nxt = g.next_nodes()
    assert nxt[0][0] == 1 and nxt[1][0] == 3 and nxt[3][0] == 4
    assert nxt[4][0] == 5 and nxt[4][1] == 7 and nxt[5][0] == 6
    assert nxt[6][0] == 8 and nxt[8][0] == 9
    id2info = {b.id(): g.block_info(b)["id"] for b in blocks.values()}
    free = [sorted(id2info[i] for i in v) for v in g.free_blocks()]
    assert free == [[], [], [], [5], [3], [2, 6], [8, 11], [7, 13], [9, 10],
                    [12, 14]]
    alive = {n: blocks[n].initialized() for n in order}
    assert alive == {n: n in ("x", "b1", "b2") for n in order}


def test_time_profiling_report():
    dev = get_default_device()
    g = _rt.Graph()
    _fwd_bwd_graph(dev, g)
    g.set_verbosity(1)
    g.set_skip_iteration(1)
    for _ in range(4):
        g.RunGraph()
    rep = g.TimeProfiling()
    assert "Time Profiling" in rep
    assert g.iteration() == 4
