"""Optimizers and the data-parallel wrapper ``DistOpt``.

Hyper-parameters that can change over time (learning rate, momentum, ...) are
one-element device tensors produced by ``DecayScheduler`` objects, so a
recorded (and hipGraph-captured) training step picks up new values without
being re-recorded.  ``SGD.apply`` is one fused kernel per parameter.

``DistOpt`` all-reduces gradients over RCCL while back-propagation is still
running: gradients are handed to the communicator in the order autograd
produces them, small ones are packed into fusion buckets, and the parameter
update consumes the reduced buckets with the 1/world_size scaling folded in.

API parity: reference python/singa/opt.py (DecayScheduler :28-68, Optimizer
:71-171, SGD :174-333, RMSProp :336-441, AdaGrad :444-533, Adam :536-683,
DistOpt :686-1094).
"""
from . import autograd
from . import singa_wrap as singa
from . import tensor
from .tensor import Tensor


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
        ret = Tensor((1,), step.device)
        ret.set_value(self.init_value)
        return ret


class ExponentialDecay(DecayScheduler):
    """init_value * decay_rate ^ (step / decay_steps)."""

    def __init__(self, init_value, decay_steps, decay_rate, staircase=False):
        super(ExponentialDecay, self).__init__(init_value)
        self.decay_steps = decay_steps
        self.decay_rate = decay_rate
        self.staircase = staircase

    def call(self, step):
        if self.staircase:
            s = step // self.decay_steps
        else:
            s = step / self.decay_steps
        ret = Tensor((1,), s.device)
        ret.set_value(self.decay_rate)
        return self.init_value * tensor.pow(ret, s)


def _as_scheduler(v, what, non_negative=False):
    if type(v) == float or type(v) == int:
        if non_negative and v < 0.0:
            raise ValueError("Invalid {} value: {}".format(what, v))
        return Constant(v)
    if isinstance(v, DecayScheduler):
        return v
    raise TypeError("Wrong %s type" % what)


class Optimizer(object):
    """Base optimizer: step counter, learning-rate schedule, update loop."""

    def __init__(self, lr, dtype=tensor.float32):
        if type(lr) == float or type(lr) == int:
            self.lr = Constant(lr)
        elif isinstance(lr, DecayScheduler):
            self.lr = lr
        else:
            raise TypeError("Wrong learning rate type")
        self.dtype = dtype
        self.step_counter = Tensor((1,), dtype=tensor.float32)
        self.step_counter.set_value(0)
        self.lr_value = self.lr(self.step_counter)

    def _state_key(self, name, param):
        """Key of a parameter's optimizer state.  It is the parameter's name;
        when two different parameter tensors carry the same name (layers used
        outside a Model are not given hierarchical names) later ones get a
        numbered key instead of silently sharing state."""
        owners = self.__dict__.setdefault("_key_owner", {})
        key, n = name, 0
        while key in owners and owners[key] != id(param):
            n += 1
            key = "%s#%d" % (name, n)
        owners[key] = id(param)
        return key

    def get_states(self):
        return {'step_counter': tensor.to_numpy(self.step_counter)[0]}

    def set_states(self, states):
        self.step_counter = Tensor((1,))
        self.step_counter.set_value(states['step_counter'])
        self.lr_value = self.lr(self.step_counter)

    def __call__(self, loss):
        self.call(loss)
        self.step()

    def call(self, loss):
        for p, g in autograd.backward(loss):
            if p.name is None:
                p.name = id(p)
            self.apply(p.name, p, g)

    def step(self):
        """Advance the step counter and refresh the scheduled values."""
        self.step_counter.data += 1
        lr_value = self.lr(self.step_counter)
        self.lr_value.copy_from(lr_value)

    def apply(self, param_name, param_value, param_grad):
        raise NotImplementedError

    def update(self, param, grad):
        """Deprecated spelling of ``apply``."""
        if param.name is None:
            param.name = id(param)
        self.apply(param.name, param, grad)

    def device_check(self, *inputs):
        dev = inputs[0].device
        flag = dev.graph_enabled()
        dev.EnableGraph(False)
        x_dev_id = dev.id()
        for var in inputs:
            if var.device.id() != x_dev_id:
                var.to_device(dev)
        dev.EnableGraph(flag)

    def backward_and_update(self, loss):
        """Deprecated spelling of ``__call__``."""
        self.__call__(loss)

    def _new_state(self, like, copy=False, dtype=None):
        """A state tensor shaped like ``like`` (zeros, or a copy of it),
        created immediately even while a graph is being recorded."""
        dev = like.device
        flag = dev.graph_enabled()
        dev.EnableGraph(False)
        if copy:
            t = like.as_type(dtype) if dtype not in (None, like.dtype) \
                else like.clone()
        elif dtype in (None, like.dtype):
            t = tensor.zeros_like(like)
        else:
            t = tensor.Tensor(like.shape, dev, dtype)
            t.set_value(0.0)
        dev.EnableGraph(flag)
        return t


class SGD(Optimizer):
    """Stochastic gradient descent with momentum / dampening / Nesterov and
    L2 weight decay:

        g <- g + weight_decay * p
        v <- momentum * v + (1 - dampening) * g
        g <- g + momentum * v   (nesterov)   |   g <- v   (plain momentum)
        p <- p - lr * g
    """

    def __init__(self, lr=0.1, momentum=0, dampening=0, weight_decay=0,
                 nesterov=False, dtype=tensor.float32, master_weights=True):
        super(SGD, self).__init__(lr, dtype)
        # 16-bit parameters updated through ``apply_many`` keep a float32
        # master copy and float32 momentum (the reference steps the half
        # tensors directly, python/singa/opt.py:259-310, and loses updates
        # below half a unit in the last place)
        self.master_weights = master_weights
        self.masters = dict()
        self.momentum = _as_scheduler(momentum, "momentum", True)
        self.mom_value = self.momentum(self.step_counter).as_type(self.dtype)
        self.dampening = _as_scheduler(dampening, "dampening")
        self.dam_value = self.dampening(self.step_counter).as_type(self.dtype)
        self.weight_decay = _as_scheduler(weight_decay, "weight_decay", True)
        self.decay_value = self.weight_decay(self.step_counter).as_type(
            self.dtype)
        self.nesterov = nesterov
        self.moments = dict()
        if nesterov and (self.momentum.init_value <= 0 or
                         self.dampening.init_value != 0):
            raise ValueError(
                "Nesterov momentum requires a momentum and zero dampening")

    def apply(self, param_name, param_value, param_grad, scale=1.0):
        assert param_value.shape == param_grad.shape, ("shape mismatch",
                                                       param_value.shape,
                                                       param_grad.shape)
        self.device_check(param_value, self.step_counter, self.lr_value,
                          self.mom_value, self.dam_value, self.decay_value)
        buf = None
        param_name = self._state_key(param_name, param_value)
        if self.momentum.init_value != 0:
            if param_name not in self.moments:
                self.moments[param_name] = self._new_state(param_value)
            buf = self.moments[param_name].data
        singa.SGDUpdate(param_value.data, param_grad.data, buf,
                        self.lr_value.data, self.mom_value.data,
                        self.dam_value.data, self.decay_value.data,
                        self.nesterov, scale)

    def apply_many(self, pairs, scale=1.0):
        """Update every (param, grad) pair with one multi-tensor kernel per
        48 parameters instead of one launch each (ResNet-50 has 161)."""
        pairs = list(pairs)
        if not pairs:
            return
        use_mom = self.momentum.init_value != 0
        groups = {}
        for p, g in pairs:
            assert p.shape == g.shape, ("shape mismatch", p.shape, g.shape)
            if p.name is None:
                p.name = id(p)
            groups.setdefault((p.dtype, id(p.device)), []).append((p, g))
        for (dt, _), items in groups.items():
            # mixed precision: fp16 weights next to fp32 BatchNorm parameters
            # are updated by one launch per dtype
            self.device_check(items[0][0], self.step_counter, self.lr_value,
                              self.mom_value, self.dam_value,
                              self.decay_value)
            mixed = self.master_weights and dt != tensor.float32
            bufs, masters = None, None
            if mixed:
                masters = []
                for p, _ in items:
                    key = self._state_key(p.name, p)
                    if key not in self.masters:
                        self.masters[key] = self._new_state(p, copy=True,
                                                            dtype=tensor.float32)
                    masters.append(self.masters[key].data)
            if use_mom:
                bufs = []
                for p, _ in items:
                    key = self._state_key(p.name, p)
                    if key not in self.moments:
                        self.moments[key] = self._new_state(
                            p, dtype=tensor.float32 if mixed else None)
                    bufs.append(self.moments[key].data)
            singa.SGDUpdateMulti([p.data for p, _ in items],
                                 [g.data for _, g in items], bufs,
                                 self.lr_value.data, self.mom_value.data,
                                 self.dam_value.data, self.decay_value.data,
                                 self.nesterov, scale, masters)

    def call(self, loss):
        """Back-propagate, then update all parameters in one fused step."""
        self.apply_many(list(autograd.backward(loss)))

    def step(self):
        super().step()
        self.mom_value.copy_from(
            self.momentum(self.step_counter).as_type(self.dtype))
        self.dam_value.copy_from(
            self.dampening(self.step_counter).as_type(self.dtype))
        self.decay_value.copy_from(
            self.weight_decay(self.step_counter).as_type(self.dtype))

    def get_states(self):
        states = super().get_states()
        if self.momentum.init_value > 0:
            states['moments'] = self.moments
        if self.masters:
            states['masters'] = self.masters
        return states

    def set_states(self, states):
        super().set_states(states)
        if 'moments' in states:
            self.moments = states['moments']
            self.mom_value = self.momentum(self.step_counter)
        if 'masters' in states:
            self.masters = states['masters']

    def forget_masters(self):
        """Drop the float32 master copies: the next step re-creates them from
        the current parameter values.  Call after parameters were overwritten
        from outside the optimizer (loading a checkpoint)."""
        self.masters = dict()


class RMSProp(Optimizer):
    """r <- rho * r + (1 - rho) * g^2 ; p <- p - lr * g / sqrt(r + eps)."""

    def __init__(self, lr=0.1, rho=0.9, epsilon=1e-8, weight_decay=0):
        super(RMSProp, self).__init__(lr)
        self.weight_decay = _as_scheduler(weight_decay, "weight_decay", True)
        self.decay_value = self.weight_decay(self.step_counter)
        self.rho = _as_scheduler(rho, "rho")
        self.rho_value = self.rho(self.step_counter)
        self.epsilon = _as_scheduler(epsilon, "epsilon")
        self.epsilon_value = self.epsilon(self.step_counter)
        self.running_average = dict()

    def apply(self, param_name, param_value, param_grad):
        assert param_value.shape == param_grad.shape, ("shape mismatch",
                                                       param_value.shape,
                                                       param_grad.shape)
        self.device_check(param_value, self.step_counter, self.lr_value,
                          self.rho_value, self.epsilon_value, self.decay_value)
        if self.weight_decay.init_value != 0:
            singa.Axpy(self.decay_value.data, param_value.data, param_grad.data)
        param_name = self._state_key(param_name, param_value)
        if param_name not in self.running_average:
            self.running_average[param_name] = self._new_state(param_value)
        ra = self.running_average[param_name]
        ra *= self.rho_value
        tmp1 = singa.Square(param_grad.data)
        tmp2 = 1.0 - self.rho_value
        singa.Axpy(tmp2.data, tmp1, ra.data)
        minus_lr = 0.0 - self.lr_value
        tmp3 = ra + self.epsilon_value
        tmp3 = singa.Sqrt(tmp3.data)
        tmp3 = singa.__div__(param_grad.data, tmp3)
        singa.Axpy(minus_lr.data, tmp3, param_value.data)

    def step(self):
        super().step()
        self.decay_value.copy_from(self.weight_decay(self.step_counter))
        self.rho_value.copy_from(self.rho(self.step_counter))
        self.epsilon_value.copy_from(self.epsilon(self.step_counter))

    def get_states(self):
        states = super().get_states()
        states['running_average'] = self.running_average
        return states

    def set_states(self, states):
        super().set_states(states)
        if 'running_average' in states:
            self.running_average = states['running_average']


class AdaGrad(Optimizer):
    """h <- h + g^2 ; p <- p - lr * g / sqrt(h + eps)."""

    def __init__(self, lr=0.1, epsilon=1e-8, weight_decay=0):
        super(AdaGrad, self).__init__(lr)
        self.weight_decay = _as_scheduler(weight_decay, "weight_decay", True)
        self.decay_value = self.weight_decay(self.step_counter)
        self.epsilon = _as_scheduler(epsilon, "epsilon")
        self.epsilon_value = self.epsilon(self.step_counter)
        self.history = dict()

    def apply(self, param_name, param_value, param_grad):
        assert param_value.shape == param_grad.shape, ("shape mismatch",
                                                       param_value.shape,
                                                       param_grad.shape)
        self.device_check(param_value, self.step_counter, self.lr_value,
                          self.epsilon_value, self.decay_value)
        if self.weight_decay.init_value != 0:
            singa.Axpy(self.decay_value.data, param_value.data, param_grad.data)
        param_name = self._state_key(param_name, param_value)
        if param_name not in self.history:
            self.history[param_name] = self._new_state(param_value)
        hist = self.history[param_name]
        hist.data += singa.Square(param_grad.data)
        minus_lr = 0.0 - self.lr_value
        tmp = hist + self.epsilon_value
        tmp = singa.Sqrt(tmp.data)
        tmp = singa.__div__(param_grad.data, tmp)
        singa.Axpy(minus_lr.data, tmp, param_value.data)

    def step(self):
        super().step()
        self.decay_value.copy_from(self.weight_decay(self.step_counter))
        self.epsilon_value.copy_from(self.epsilon(self.step_counter))

    def get_states(self):
        states = super().get_states()
        states['history'] = self.history
        return states

    def set_states(self, states):
        super().set_states(states)
        if 'history' in states:
            self.history = states['history']


class Adam(Optimizer):
    """Adam with bias correction:

        m <- b1 * m + (1 - b1) * g ;  v <- b2 * v + (1 - b2) * g^2
        p <- p - lr * (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
    """

    def __init__(self, lr=0.1, beta_1=0.9, beta_2=0.999, epsilon=1e-8,
                 weight_decay=0):
        super(Adam, self).__init__(lr)
        self.weight_decay = _as_scheduler(weight_decay, "weight_decay", True)
        self.decay_value = self.weight_decay(self.step_counter)
        self.beta_1 = _as_scheduler(beta_1, "beta_1")
        self.beta_1_value = self.beta_1(self.step_counter)
        self.beta_2 = _as_scheduler(beta_2, "beta_2")
        self.beta_2_value = self.beta_2(self.step_counter)
        self.epsilon = _as_scheduler(epsilon, "epsilon")
        self.epsilon_value = self.epsilon(self.step_counter)
        self.m = dict()
        self.v = dict()

    def apply(self, param_name, param_value, param_grad):
        assert param_value.shape == param_grad.shape, ("shape mismatch",
                                                       param_value.shape,
                                                       param_grad.shape)
        self.device_check(param_value, self.step_counter, self.lr_value,
                          self.beta_1_value, self.beta_2_value,
                          self.epsilon_value, self.decay_value)
        if self.weight_decay.init_value != 0:
            singa.Axpy(self.decay_value.data, param_value.data, param_grad.data)
        param_name = self._state_key(param_name, param_value)
        if param_name not in self.m:
            self.m[param_name] = self._new_state(param_value)
            self.v[param_name] = self._new_state(param_value)
        step = self.step_counter + 1.0

        tmp = 1.0 - self.beta_1_value
        self.m[param_name] *= self.beta_1_value
        singa.Axpy(tmp.data, param_grad.data, self.m[param_name].data)

        tmp = 1.0 - self.beta_2_value
        self.v[param_name] *= self.beta_2_value
        singa.Axpy(tmp.data, singa.Square(param_grad.data),
                   self.v[param_name].data)

        tmp = 1.0 - tensor.pow(self.beta_1_value, step)
        m_norm = self.m[param_name] / tmp
        tmp = 1.0 - tensor.pow(self.beta_2_value, step)
        v_norm = self.v[param_name] / tmp

        a = tensor.sqrt(v_norm) + self.epsilon_value
        tmp = m_norm / a
        minus_lr = 0.0 - self.lr_value
        singa.Axpy(minus_lr.data, tmp.data, param_value.data)

    def step(self):
        super().step()
        self.decay_value.copy_from(self.weight_decay(self.step_counter))
        self.beta_1_value.copy_from(self.beta_1(self.step_counter))
        self.beta_2_value.copy_from(self.beta_2(self.step_counter))

    def get_states(self):
        states = super().get_states()
        states['m'] = self.m
        states['v'] = self.v
        return states

    def set_states(self, states):
        super().set_states(states)
        if 'm' in states:
            self.m = states['m']
        if 'v' in states:
            self.v = states['v']


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
        if nccl_id is None:
            self.communicator = singa.Communicator(buffSize)
        else:
            self.communicator = singa.Communicator(local_rank, world_size,
                                                   nccl_id, buffSize)
        self.world_size = self.communicator.world_size
        self.local_rank = self.communicator.local_rank
        self.global_rank = self.communicator.global_rank
        # elements per fusion bucket when the caller does not pass one; ring
        # all-reduce over xGMI favours fewer, larger buckets than the
        # reference's 2M-element default
        self.threshold = 2097152

    def __call__(self, loss):
        self.backward_and_update(loss, self.threshold)

    def update(self, param, grad):
        """Average the all-reduced gradient and apply the wrapped optimizer.
        For SGD the division is folded into the fused update kernel."""
        if isinstance(self.opt, SGD):
            if param.name is None:
                param.name = id(param)
            self.opt.apply(param.name, param, grad,
                           scale=1.0 / self.world_size)
        else:
            grad /= self.world_size
            self.opt.update(param, grad)

    def update_many(self, pairs):
        """``update`` for a list of (param, grad) pairs; SGD takes the
        multi-tensor path with the averaging folded in."""
        if isinstance(self.opt, SGD):
            self.opt.apply_many(pairs, scale=1.0 / self.world_size)
        else:
            for p, g in pairs:
                self.update(p, g)

    def all_reduce(self, tensor):
        self.communicator.synch(tensor)

    def fused_all_reduce(self, tensor, send=True):
        self.communicator.fusedSynch(singa.VecTensor(tensor), send)

    def all_reduce_half(self, tensor):
        self.communicator.synchHalf(tensor)

    def fused_all_reduce_half(self, tensor, send=True):
        self.communicator.fusedSynchHalf(singa.VecTensor(tensor), send)

    def sparsification(self, tensor, accumulation, spars, topK):
        if accumulation is None:
            self.communicator.sparsification(tensor, spars, topK)
        else:
            self.communicator.sparsification(tensor, accumulation, spars, topK)

    def fused_sparsification(self, tensor, accumulation, spars, topK):
        tensor = singa.VecTensor(tensor)
        if accumulation is None:
            self.communicator.fusedSparsification(tensor, spars, topK)
        else:
            self.communicator.fusedSparsification(tensor, accumulation, spars,
                                                  topK)

    def wait(self):
        self.communicator.wait()

    def _reduce_loop(self, loss, threshold, reduce_one, reduce_fused,
                     pre=None):
        """Shared gradient schedule: big gradients are reduced on their own,
        small ones are packed until the bucket exceeds ``threshold`` elements.
        A bucket is a flat buffer of one element type, so gradients are
        bucketed per dtype (16-bit weight gradients and the float32
        batch-norm gradients of a mixed-precision model fill separate
        buckets)."""
        plist = []
        pending = {}          # dtype -> [element count, gradients]
        for p, g in autograd.backward(loss):
            if pre is not None:
                g = pre(p, g)
            if g.size() > threshold:
                reduce_one(g.data)
            else:
                slot = pending.setdefault(g.dtype, [0, []])
                slot[1].append(g.data)
                reduce_fused([g.data], send=False)
                slot[0] += g.size()
                if slot[0] > threshold:
                    reduce_fused(slot[1])
                    del pending[g.dtype]
            plist.append((p, g))
        for _, glist in pending.values():
            reduce_fused(glist)
        self.wait()
        self.update_many(plist)
        self.opt.step()

    def backward_and_update(self, loss, threshold=2097152):
        """Back-propagate, all-reduce (fp32) and update."""
        self._reduce_loop(loss, threshold, self.all_reduce,
                          self.fused_all_reduce)

    def backward_and_update_half(self, loss, threshold=2097152, clipping=False,
                                 clip_Value=100):
        """Same schedule with gradients converted to fp16 on the wire."""

        def pre(p, g):
            assert p.dtype == tensor.float32, (
                'This function is only available for input tensor precision 32 bit, '
                'which are converted into 16 bits before transmit')
            if clipping:
                g = autograd.clip(g, -clip_Value, clip_Value)
            return g

        self._reduce_loop(loss, threshold, self.all_reduce_half,
                          self.fused_all_reduce_half, pre)

    def backward_and_partial_update(self, loss, threshold=2097152):
        """Every rank updates locally; each iteration one partition of the
        *parameters* (round-robin) is all-reduced and averaged."""
        if not hasattr(self, "partial"):
            self.partial = 0
        self.partial += 1
        k = 0
        plist = []
        acc = 0
        tenlist = []
        reduced = []
        for p, g in autograd.backward(loss):
            self.opt.update(p, g)
            if p.size() > threshold:
                k += 1
                if k == self.partial:
                    self.all_reduce(p.data)
                    reduced.append(p)
            else:
                plist.append(p.data)
                tenlist.append(p)
                acc += p.size()
                if acc > threshold:
                    k += 1
                    if k == self.partial:
                        self.fused_all_reduce(plist, send=False)
                        self.fused_all_reduce(plist)
                        reduced = tenlist
                    acc = 0
                    plist = []
                    tenlist = []
        if plist:
            k += 1
            if k == self.partial:
                self.fused_all_reduce(plist, send=False)
                self.fused_all_reduce(plist)
                reduced = tenlist
        self.wait()
        for r in reduced:
            r /= self.world_size
        if k == self.partial:
            self.partial = 0
        self.opt.step()

    def backward_and_sparse_update(self, loss, threshold=2097152, spars=0.05,
                                   topK=False, corr=True):
        """Sparsified gradient exchange (threshold or top-K) with optional
        local residual accumulation (error feedback)."""
        if (not hasattr(self, "sparsInit")) and corr:
            self.gradAccumulation = []
            self.sparsInit = False
        plist = []
        acc = 0
        k = -1
        glist = []

        def residual(p, n):
            if corr and not self.sparsInit:
                flag = p.device.graph_enabled()
                p.device.EnableGraph(False)
                t = tensor.Tensor((n,), p.device, p.dtype)
                t.set_value(0.0)
                self.gradAccumulation.append(t)
                p.device.EnableGraph(flag)

        for p, g in autograd.backward(loss):
            if g.size() > threshold:
                k += 1
                residual(p, g.size())
                self.sparsification(
                    g.data, self.gradAccumulation[k].data if corr else None,
                    spars, topK)
            else:
                glist.append(g.data)
                acc += g.size()
                if acc > threshold:
                    k += 1
                    residual(p, acc)
                    self.fused_sparsification(
                        glist,
                        self.gradAccumulation[k].data if corr else None,
                        spars, topK)
                    acc = 0
                    glist = []
            plist.append((p, g))
        if glist:
            k += 1
            residual(p, acc)
            self.fused_sparsification(
                glist, self.gradAccumulation[k].data if corr else None, spars,
                topK)
        self.wait()
        for p, g in plist:
            self.update(p, g)
        self.sparsInit = True
        self.opt.step()
