"""Optimizers and the data-parallel wrapper ``DistOpt``.

Everything a recorded training step reads lives in device tensors that are
updated *in place*: the step counter, every scheduled hyper-parameter (learning
rate, momentum, ...), the per-parameter state and the float32 master copies of
16-bit weights.  A step captured into a hipGraph therefore follows schedules
and restored checkpoints without being recorded again.

Every rule is ONE fused multi-tensor kernel per dtype group (gradient scaling
for data-parallel averaging, weight decay, moment updates and the parameter
step in a single pass over the tensors), not a chain of axpy / scale launches
per parameter.

``DistOpt`` all-reduces gradients over RCCL while back-propagation is still
running: gradients reach the communicator in the order autograd completes
them, small ones travel in fusion buckets (one per dtype), and the fused
update consumes the reduced gradients with the 1 / world_size factor folded in.

API parity: reference python/singa/opt.py (DecayScheduler :28-68, Optimizer
:71-171, SGD :174-333, RMSProp :336-441, AdaGrad :444-533, Adam :536-683,
DistOpt :686-1094).
"""
from collections import OrderedDict

from . import autograd
from . import singa_wrap as singa
from . import tensor
from .tensor import Tensor


# ---------------------------------------------------------------------------
# schedules
# ---------------------------------------------------------------------------
class DecayScheduler:
    """Maps the step counter (a tensor) to the current value (a tensor)."""

    def __init__(self, init_value):
        self.init_value = init_value

    def __call__(self, step):
        assert isinstance(step, Tensor)
        return self.call(step)

    def call(self, step) -> Tensor:
        raise NotImplementedError


class Constant(DecayScheduler):

    def call(self, step: Tensor) -> Tensor:
        value = Tensor((1,), step.device)
        value.set_value(self.init_value)
        return value


class ExponentialDecay(DecayScheduler):
    """init_value * decay_rate ^ (step / decay_steps); with ``staircase`` the
    exponent is the integer part."""

    def __init__(self, init_value, decay_steps, decay_rate, staircase=False):
        super(ExponentialDecay, self).__init__(init_value)
        self.decay_steps, self.decay_rate = decay_steps, decay_rate
        self.staircase = bool(staircase)

    def call(self, step):
        exponent = (step // self.decay_steps) if self.staircase \
            else (step / self.decay_steps)
        base = Tensor((1,), exponent.device)
        base.set_value(self.decay_rate)
        return self.init_value * tensor.pow(base, exponent)


def _schedule(spec, what, non_negative=False):
    """A number becomes a constant schedule; schedules pass through."""
    if isinstance(spec, DecayScheduler):
        return spec
    if isinstance(spec, (int, float)) and not isinstance(spec, bool):
        if non_negative and spec < 0.0:
            raise ValueError("Invalid {} value: {}".format(what, spec))
        return Constant(spec)
    raise TypeError("Wrong %s type" % what)


def _eagerly(dev, fn):
    """Run ``fn`` with graph recording off on ``dev`` (state creation must not
    become part of a recorded step)."""
    recording = dev.graph_enabled()
    dev.EnableGraph(False)
    try:
        return fn()
    finally:
        dev.EnableGraph(recording)


def _name_of(p):
    if p.name is None:
        p.name = id(p)
    return p.name


# ---------------------------------------------------------------------------
# base class
# ---------------------------------------------------------------------------
class Optimizer(object):
    """Step counter, scheduled hyper-parameters and per-parameter state.

    A subclass registers its scheduled scalars with ``_hyper(attr, spec,
    value_attr)``: the schedule is stored as ``self.<attr>`` and its current
    value as the device tensor ``self.<value_attr>``, which ``step()``
    refreshes in place."""

    def __init__(self, lr, dtype=tensor.float32):
        self.dtype = dtype
        self.step_counter = Tensor((1,), dtype=tensor.float32)
        self.step_counter.set_value(0)
        self._scheduled = []          # (schedule attr, value attr, cast?)
        self._key_owner = {}          # state key -> id(parameter)
        self._hyper("lr", lr, "lr_value", what="learning rate")

    # ---- hyper-parameters ---------------------------------------------------------
    def _hyper(self, attr, spec, value_attr, what=None, non_negative=False,
               cast=False):
        sched = _schedule(spec, what or attr, non_negative)
        setattr(self, attr, sched)
        setattr(self, value_attr, self._evaluate(sched, cast))
        self._scheduled.append((attr, value_attr, cast))
        return sched

    def _evaluate(self, sched, cast):
        v = sched(self.step_counter)
        return v.as_type(self.dtype) if cast else v

    def _values(self):
        return [getattr(self, v) for _, v, _ in self._scheduled]

    def _refresh(self):
        for attr, value_attr, cast in self._scheduled:
            getattr(self, value_attr).copy_from(
                self._evaluate(getattr(self, attr), cast))

    def step(self):
        """Advance the step counter and refresh every scheduled value."""
        self.step_counter.data += 1
        self._refresh()

    # ---- per-parameter state -------------------------------------------------------
    def _state_key(self, name, param):
        """Key of a parameter's state: its name, or a numbered variant when a
        different tensor already owns that name (layers used outside a Model
        carry no hierarchical names and would otherwise share state)."""
        key, n = name, 0
        while self._key_owner.get(key, id(param)) != id(param):
            n += 1
            key = "%s#%d" % (name, n)
        self._key_owner[key] = id(param)
        return key

    def _slot(self, table, key, like, dtype=None, copy=False):
        """State tensor ``table[key]`` shaped like ``like`` (zeros, or a copy
        of it), created eagerly on first use."""
        t = table.get(key)
        if t is None:
            def make():
                if copy:
                    return like.as_type(dtype) if dtype not in (
                        None, like.dtype) else like.clone()
                s = Tensor(like.shape, like.device,
                           like.dtype if dtype is None else dtype)
                s.set_value(0.0)
                return s
            t = table[key] = _eagerly(like.device, make)
        else:
            # a slot adopted from a checkpoint (set_states before the first
            # step) was created wherever the file was read -- the host -- and
            # in the type it was saved in: it follows the parameter before a
            # kernel is handed its address
            want = like.dtype if dtype is None else dtype
            if t.device.id() != like.device.id() or t.dtype != want:
                def adopt():
                    moved = t if t.dtype == want else t.as_type(want)
                    if moved.device.id() != like.device.id():
                        moved.to_device(like.device)
                    return moved
                t = table[key] = _eagerly(like.device, adopt)
        return t

    @staticmethod
    def _restore(table, saved):
        """Load saved state tensors: existing ones are overwritten in place
        (a recorded step keeps reading them), new ones are adopted."""
        for key, value in saved.items():
            mine = table.get(key)
            if mine is not None and mine is not value and \
                    mine.shape == value.shape:
                mine.copy_from(value if value.dtype == mine.dtype
                               else value.as_type(mine.dtype))
            else:
                table[key] = value

    def device_check(self, *inputs):
        """Move every tensor to the device of the first (never recorded)."""
        dev = inputs[0].device

        def move():
            for t in inputs[1:]:
                if t.device.id() != dev.id():
                    t.to_device(dev)
        _eagerly(dev, move)

    # ---- update entry points ---------------------------------------------------------
    def __call__(self, loss):
        self.call(loss)
        self.step()

    def call(self, loss):
        """Back-propagate, then update all parameters in one fused step."""
        self.apply_many(list(autograd.backward(loss)))

    def backward_and_update(self, loss):
        """Deprecated spelling of ``__call__``."""
        self.__call__(loss)

    def apply(self, param_name, param_value, param_grad, scale=1.0):
        """Update one parameter from its gradient (``scale`` multiplies the
        gradient first)."""
        assert param_value.shape == param_grad.shape, (
            "shape mismatch", param_value.shape, param_grad.shape)
        self._update([(param_name, param_value, param_grad)], scale)

    def update(self, param, grad):
        """Deprecated spelling of ``apply``."""
        self.apply(_name_of(param), param, grad)

    def apply_many(self, pairs, scale=1.0):
        """Update every (param, grad) pair; parameters of one dtype on one
        device share a launch (a mixed-precision model has 16-bit weights next
        to float32 batch-norm parameters)."""
        groups = OrderedDict()
        for p, g in pairs:
            assert p.shape == g.shape, ("shape mismatch", p.shape, g.shape)
            groups.setdefault((p.dtype, id(p.device)), []).append(
                (_name_of(p), p, g))
        for group in groups.values():
            self._update(group, scale)

    def _update(self, items, scale):
        """items: [(name, param, grad)] of one dtype on one device."""
        raise NotImplementedError

    # ---- checkpointing -----------------------------------------------------------------
    def get_states(self):
        return {'step_counter': tensor.to_numpy(self.step_counter)[0]}

    def set_states(self, states):
        self.step_counter.set_value(float(states['step_counter']))
        self._refresh()


# ---------------------------------------------------------------------------
# SGD
# ---------------------------------------------------------------------------
class SGD(Optimizer):
    """Stochastic gradient descent with momentum / dampening / Nesterov and
    L2 weight decay:

        g <- g + weight_decay * p
        v <- momentum * v + (1 - dampening) * g
        g <- g + momentum * v   (nesterov)   |   g <- v   (plain momentum)
        p <- p - lr * g

    16-bit parameters keep a float32 master copy and float32 momentum
    (``master_weights``): the reference steps the half tensors directly
    (python/singa/opt.py:259-310) and loses updates below half a unit in the
    last place."""

    def __init__(self, lr=0.1, momentum=0, dampening=0, weight_decay=0,
                 nesterov=False, dtype=tensor.float32, master_weights=True):
        super(SGD, self).__init__(lr, dtype)
        self.master_weights = master_weights
        self._hyper("momentum", momentum, "mom_value", non_negative=True,
                    cast=True)
        self._hyper("dampening", dampening, "dam_value", cast=True)
        self._hyper("weight_decay", weight_decay, "decay_value",
                    non_negative=True, cast=True)
        self.nesterov = nesterov
        if nesterov and (self.momentum.init_value <= 0 or
                         self.dampening.init_value != 0):
            raise ValueError(
                "Nesterov momentum requires a momentum and zero dampening")
        self.moments = dict()
        self.masters = dict()

    def _update(self, items, scale):
        lead = items[0][1]
        self.device_check(lead, self.step_counter, *self._values())
        mixed = self.master_weights and lead.dtype != tensor.float32 and \
            len(items) > 0
        keys = [self._state_key(n, p) for n, p, _ in items]
        masters = None
        if mixed:
            masters = [self._slot(self.masters, k, p, tensor.float32,
                                  copy=True).data
                       for k, (_, p, _) in zip(keys, items)]
        bufs = None
        if self.momentum.init_value != 0:
            bufs = [self._slot(self.moments, k, p,
                               tensor.float32 if mixed else None).data
                    for k, (_, p, _) in zip(keys, items)]
        hyp = (self.lr_value.data, self.mom_value.data, self.dam_value.data,
               self.decay_value.data)
        singa.SGDUpdateMulti([p.data for _, p, _ in items],
                             [g.data for _, _, g in items], bufs, *hyp,
                             self.nesterov, scale, masters)

    def sync_masters(self, params):
        """Overwrite the float32 master copies with the current parameter
        values, in place (the parameters were loaded from a checkpoint; a
        recorded step keeps reading the same master tensors).
        ``params``: name -> Tensor."""
        key_of = {owner: key for key, owner in self._key_owner.items()
                  if key in self.masters}
        for p in params.values():
            key = key_of.get(id(p))
            if key is not None:
                self.masters[key].copy_from(p.as_type(tensor.float32))

    def forget_masters(self):
        """Drop the master copies; the next step re-creates them from the
        parameters.  Only meaningful before the step has been recorded."""
        self.masters = dict()

    def get_states(self):
        states = super(SGD, self).get_states()
        if self.momentum.init_value > 0:
            states['moments'] = self.moments
        if self.masters:
            states['masters'] = self.masters
        return states

    def set_states(self, states):
        super(SGD, self).set_states(states)
        self._restore(self.moments, states.get('moments', {}))
        self._restore(self.masters, states.get('masters', {}))


# ---------------------------------------------------------------------------
# adaptive rules (one fused kernel each)
# ---------------------------------------------------------------------------
class _Adaptive(Optimizer):
    """Shared driver of RMSProp / AdaGrad / Adam.  A subclass names its kernel
    rule, its one or two state tables (``_tables``) and the schedules feeding
    the kernel's two rule-specific scalars (``_ab``)."""

    rule = None
    uses_step = False
    state_names = ()

    def _tables(self):
        return [getattr(self, n) for n in self.state_names]

    def _ab(self):
        return None, None

    def _update(self, items, scale):
        lead = items[0][1]
        self.device_check(lead, self.step_counter, *self._values())
        keys = [self._state_key(n, p) for n, p, _ in items]
        states = [[self._slot(tb, k, p).data
                   for k, (_, p, _) in zip(keys, items)]
                  for tb in self._tables()]
        a, b = self._ab()
        hypers = (self.lr_value.data, self.decay_value.data,
                  self.epsilon_value.data, a, b,
                  self.step_counter.data if self.uses_step else None)
        singa.AdaptiveUpdateMulti(
            self.rule, [p.data for _, p, _ in items],
            [g.data for _, _, g in items], states[0],
            states[1] if len(states) > 1 else None, hypers, scale)

    def get_states(self):
        states = super(_Adaptive, self).get_states()
        for n in self.state_names:
            states[n] = getattr(self, n)
        return states

    def set_states(self, states):
        super(_Adaptive, self).set_states(states)
        for n in self.state_names:
            self._restore(getattr(self, n), states.get(n, {}))


class RMSProp(_Adaptive):
    """r <- rho * r + (1 - rho) * g^2 ; p <- p - lr * g / sqrt(r + eps)."""

    rule = singa.ADAPT_RMSPROP
    state_names = ('running_average',)

    def __init__(self, lr=0.1, rho=0.9, epsilon=1e-8, weight_decay=0):
        super(RMSProp, self).__init__(lr)
        self._hyper("weight_decay", weight_decay, "decay_value",
                    non_negative=True)
        self._hyper("rho", rho, "rho_value")
        self._hyper("epsilon", epsilon, "epsilon_value")
        self.running_average = dict()

    def _ab(self):
        return self.rho_value.data, None


class AdaGrad(_Adaptive):
    """h <- h + g^2 ; p <- p - lr * g / sqrt(h + eps)."""

    rule = singa.ADAPT_ADAGRAD
    state_names = ('history',)

    def __init__(self, lr=0.1, epsilon=1e-8, weight_decay=0):
        super(AdaGrad, self).__init__(lr)
        self._hyper("weight_decay", weight_decay, "decay_value",
                    non_negative=True)
        self._hyper("epsilon", epsilon, "epsilon_value")
        self.history = dict()


class Adam(_Adaptive):
    """Adam with bias correction (t counts from 1):

        m <- b1 * m + (1 - b1) * g ;  v <- b2 * v + (1 - b2) * g^2
        p <- p - lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
    """

    rule = singa.ADAPT_ADAM
    uses_step = True
    state_names = ('m', 'v')

    def __init__(self, lr=0.1, beta_1=0.9, beta_2=0.999, epsilon=1e-8,
                 weight_decay=0):
        super(Adam, self).__init__(lr)
        self._hyper("weight_decay", weight_decay, "decay_value",
                    non_negative=True)
        self._hyper("beta_1", beta_1, "beta_1_value")
        self._hyper("beta_2", beta_2, "beta_2_value")
        self._hyper("epsilon", epsilon, "epsilon_value")
        self.m = dict()
        self.v = dict()

    def _ab(self):
        return self.beta_1_value.data, self.beta_2_value.data


# ---------------------------------------------------------------------------
# data-parallel wrapper
# ---------------------------------------------------------------------------
class _Buckets(object):
    """Gradients waiting for a fused exchange, one open bucket per dtype (a
    flat bucket has one element type: the 16-bit weight gradients and the
    float32 batch-norm gradients of a mixed-precision model fill separate
    buckets).  ``add`` returns a bucket's members once it has grown beyond
    ``threshold`` elements."""

    def __init__(self, threshold):
        self.threshold = threshold
        self.open = OrderedDict()      # dtype -> [elements, [core tensors]]

    def add(self, t):
        slot = self.open.setdefault(t.dtype, [0, []])
        slot[0] += t.size()
        slot[1].append(t.data)
        if slot[0] > self.threshold:
            return self.open.pop(t.dtype)[1]
        return None

    def drain(self):
        rest = [members for _, members in self.open.values()]
        self.open = OrderedDict()
        return rest


class DistOpt(object):
    """Wraps an optimizer for synchronous data-parallel training, one process
    per GPU.  With ``nccl_id=None`` rank / world size come from the launcher's
    environment (torchrun or mpiexec); otherwise from the explicit arguments
    (multiprocessing launch, ``NcclIdHolder`` carries the rendezvous address).

    Attributes: ``world_size``, ``local_rank``, ``global_rank``.
    """

    def __init__(self, opt=None, nccl_id=None, local_rank=None,
                 world_size=None, buffSize=4194304):
        self.opt = opt if opt is not None else SGD()
        where = () if nccl_id is None else (local_rank, world_size, nccl_id)
        self.communicator = comm = singa.Communicator(*where, buffSize)
        self.world_size, self.local_rank, self.global_rank = (
            comm.world_size, comm.local_rank, comm.global_rank)
        # elements per fusion bucket used by ``__call__`` (the reference's
        # default; callers tuning for xGMI rings pass a larger one)
        self.threshold = 2097152

    def __call__(self, loss):
        self.backward_and_update(loss, self.threshold)

    # ---- collectives (thin names over the communicator) ------------------------------
    def all_reduce(self, tensor):
        self.communicator.synch(tensor)

    def fused_all_reduce(self, tensor, send=True):
        self.communicator.fusedSynch(singa.VecTensor(tensor), send)

    def all_reduce_half(self, tensor):
        self.communicator.synchHalf(tensor)

    def fused_all_reduce_half(self, tensor, send=True):
        self.communicator.fusedSynchHalf(singa.VecTensor(tensor), send)

    def sparsification(self, tensor, accumulation, spars, topK):
        args = (spars, topK) if accumulation is None else \
            (accumulation, spars, topK)
        self.communicator.sparsification(tensor, *args)

    def fused_sparsification(self, tensor, accumulation, spars, topK):
        args = (spars, topK) if accumulation is None else \
            (accumulation, spars, topK)
        self.communicator.fusedSparsification(singa.VecTensor(tensor), *args)

    def wait(self):
        self.communicator.wait()

    # ---- updates with the averaging folded in --------------------------------------------
    def update(self, param, grad):
        """Apply the wrapped optimizer to an all-reduced (summed) gradient:
        the 1 / world_size factor is part of the fused update."""
        if type(self.opt).apply is Optimizer.apply:
            self.opt.apply(_name_of(param), param, grad,
                           scale=1.0 / self.world_size)
        else:        # a user-defined rule that knows nothing about scaling
            grad /= self.world_size
            self.opt.apply(_name_of(param), param, grad)

    def update_many(self, pairs):
        if type(self.opt).apply is Optimizer.apply:
            self.opt.apply_many(pairs, scale=1.0 / self.world_size)
        else:
            for p, g in pairs:
                self.update(p, g)

    # ---- schedules --------------------------------------------------------------------------
    def _exchange(self, loss, threshold, single, fused, prepare=None):
        """Walk the gradients as back-propagation completes them: a gradient
        above ``threshold`` elements is exchanged on its own, the others join
        their dtype's bucket, which is exchanged once it exceeds the
        threshold.  Returns the (param, grad) pairs in arrival order."""
        pairs = []
        buckets = _Buckets(threshold)
        for p, g in autograd.backward(loss):
            if prepare is not None:
                g = prepare(p, g)
            pairs.append((p, g))
            if g.size() > threshold:
                single(g.data)
                continue
            fused([g.data], send=False)        # announce the member
            full = buckets.add(g)
            if full is not None:
                fused(full)
        for members in buckets.drain():
            fused(members)
        self.wait()
        return pairs

    def backward_and_update(self, loss, threshold=2097152):
        """Back-propagate, all-reduce in the gradients' own precision and
        update."""
        self.update_many(self._exchange(loss, threshold, self.all_reduce,
                                        self.fused_all_reduce))
        self.opt.step()

    def backward_and_update_half(self, loss, threshold=2097152, clipping=False,
                                 clip_Value=100):
        """The same schedule with fp32 gradients converted to fp16 on the
        wire (optionally clipped first)."""

        def prepare(p, g):
            return autograd.clip(g, -clip_Value, clip_Value) if clipping \
                else g

        # gradients that already are 16-bit (a model trained in fp16 / bf16)
        # travel as they are; only float32 ones are converted for the wire
        def is_f32(t):
            return t.data_type() == tensor.float32

        def single(t):
            (self.all_reduce_half if is_f32(t) else self.all_reduce)(t)

        def fused(members, send=True):
            wide = [t for t in members if is_f32(t)]
            narrow = [t for t in members if not is_f32(t)]
            if wide:
                self.fused_all_reduce_half(wide, send=send)
            if narrow:
                self.fused_all_reduce(narrow, send=send)

        self.update_many(self._exchange(loss, threshold, single, fused,
                                        prepare))
        self.opt.step()

    def backward_and_partial_update(self, loss, threshold=2097152):
        """Every rank updates from its local gradients; per iteration ONE
        partition of the *parameters* (partitions are filled like gradient
        buckets and taken round-robin) is all-reduced and averaged."""
        turn = getattr(self, "partial", 0) + 1
        index = 0
        chosen = []                 # parameters exchanged this iteration
        part, part_size = [], 0

        def close_partition():
            nonlocal index, chosen, part, part_size
            index += 1
            if index == turn:
                members = [q.data for q in part]
                self.fused_all_reduce(members, send=False)
                self.fused_all_reduce(members)
                chosen = part
            part, part_size = [], 0

        for p, g in autograd.backward(loss):
            self.opt.update(p, g)
            if p.size() > threshold:
                index += 1
                if index == turn:
                    self.all_reduce(p.data)
                    chosen = [p]
                continue
            part.append(p)
            part_size += p.size()
            if part_size > threshold:
                close_partition()
        if part:
            close_partition()
        self.wait()
        for p in chosen:
            p /= self.world_size
        # 16-bit parameters are stepped from float32 master copies: the
        # averaged values must reach those, or the next step overwrites them
        refresh = getattr(self.opt, "sync_masters", None)
        if refresh is not None and chosen:
            refresh(dict(enumerate(chosen)))
        self.partial = 0 if index == turn else turn
        self.opt.step()

    def backward_and_sparse_update(self, loss, threshold=2097152, spars=0.05,
                                   topK=False, corr=True):
        """Sparsified gradient exchange (entries above a magnitude threshold,
        or the top ``spars`` fraction) with optional error feedback: what a
        rank does not send stays in a residual that is added next time."""
        first_pass = not getattr(self, "sparsInit", False)
        if corr and first_pass:
            self.gradAccumulation = []
        exchanges = [0]

        def residual(like, n):
            """The accumulator of the next exchange, created on first use."""
            k = exchanges[0]
            exchanges[0] += 1
            if not corr:
                return None
            if first_pass:
                def make():
                    t = Tensor((n,), like.device, like.dtype)
                    t.set_value(0.0)
                    return t
                self.gradAccumulation.append(_eagerly(like.device, make))
            return self.gradAccumulation[k].data

        pairs = []
        members, count = [], 0
        for p, g in autograd.backward(loss):
            pairs.append((p, g))
            if g.size() > threshold:
                self.sparsification(g.data, residual(p, g.size()), spars, topK)
                continue
            members.append(g.data)
            count += g.size()
            if count > threshold:
                self.fused_sparsification(members, residual(p, count), spars,
                                          topK)
                members, count = [], 0
        if members:
            self.fused_sparsification(members, residual(pairs[-1][0], count),
                                      spars, topK)
        self.wait()
        for p, g in pairs:
            self.update(p, g)
        self.sparsInit = True
        self.opt.step()
