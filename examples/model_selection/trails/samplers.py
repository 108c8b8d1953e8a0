"""Architecture samplers: every sampler proposes an architecture id with
``sample_next_arch()`` and learns from ``fit_sampler(arch_id, score)``.

Role: reference src/controller/controler.py (SampleController),
sampler_all/seq_sampler.py, sampler_ea/regularized_ea.py (aging evolution),
sampler_bohb/bohb_or.py (density-ratio model over configurations); a
REINFORCE policy is added."""
import collections

import numpy as np


class Sampler(object):

    def __init__(self, space, seed=0):
        self.space = space
        self.rng = np.random.RandomState(seed)
        self.history = collections.OrderedDict()

    def sample_next_arch(self):
        raise NotImplementedError

    def fit_sampler(self, arch_id, score):
        self.history[arch_id] = float(score)

    def best(self, k=1):
        ranked = sorted(self.history.items(), key=lambda t: -t[1])
        return ranked[:k]


class SequenceSampler(Sampler):
    """Every architecture of the space once, in enumeration order."""

    def __init__(self, space, seed=0):
        super(SequenceSampler, self).__init__(space, seed)
        self.stream = space.sample_all_models()

    def sample_next_arch(self):
        return next(self.stream, None)


class RandomSampler(Sampler):
    """Uniform over the space, without repetition while that is possible."""

    def sample_next_arch(self):
        for _ in range(64):
            a = self.space.random_architecture_id(self.rng)
            if a not in self.history:
                return a
        return a


class RegularizedEASampler(Sampler):
    """Aging evolution: a bounded population; a proposal mutates the best of
    a random sample of it; once full the OLDEST member is retired, whatever
    its score."""

    def __init__(self, space, population_size=10, sample_size=3, seed=0):
        super(RegularizedEASampler, self).__init__(space, seed)
        self.population = collections.deque()
        self.population_size, self.sample_size = population_size, sample_size

    def sample_next_arch(self):
        if len(self.population) < self.population_size:
            return self.space.random_architecture_id(self.rng)
        picks = self.rng.choice(len(self.population), self.sample_size,
                                replace=False)
        parent = max((self.population[i] for i in picks),
                     key=lambda t: t[1])[0]
        return self.space.mutate_architecture(parent, self.rng)

    def fit_sampler(self, arch_id, score):
        super(RegularizedEASampler, self).fit_sampler(arch_id, score)
        self.population.append((arch_id, float(score)))
        if len(self.population) > self.population_size:
            self.population.popleft()


class BohbSampler(Sampler):
    """Density-ratio proposals (the model-based half of BOHB): the observed
    configurations are split into good (top ``gamma``) and bad; per decision
    a smoothed categorical density is fitted to each group; candidates drawn
    from the good density are ranked by l(x) / g(x).  A fraction of proposals
    stays uniform."""

    def __init__(self, space, gamma=0.25, min_points=8, candidates=32,
                 random_fraction=0.2, seed=0):
        super(BohbSampler, self).__init__(space, seed)
        self.gamma, self.min_points = gamma, min_points
        self.candidates, self.random_fraction = candidates, random_fraction

    def _densities(self):
        ranked = sorted(self.history.items(), key=lambda t: -t[1])
        n_good = max(2, int(np.ceil(self.gamma * len(ranked))))
        groups = ranked[:n_good], ranked[n_good:]
        out = []
        for group in groups:
            dens = []
            for slot, options in enumerate(self.space.decisions()):
                counts = np.ones(len(options))          # add-one smoothing
                for arch_id, _ in group:
                    counts[self.space.encode(arch_id)[slot]] += 1
                dens.append(counts / counts.sum())
            out.append(dens)
        return out

    def sample_next_arch(self):
        if len(self.history) < self.min_points or \
                self.rng.rand() < self.random_fraction:
            return self.space.random_architecture_id(self.rng)
        good, bad = self._densities()
        best, best_ratio = None, -np.inf
        for _ in range(self.candidates):
            choice = [self.rng.choice(len(p), p=p) for p in good]
            ratio = sum(np.log(g[c]) - np.log(b[c])
                        for g, b, c in zip(good, bad, choice))
            arch = self.space.decode(choice)
            if ratio > best_ratio and arch not in self.history:
                best, best_ratio = arch, ratio
        return best or self.space.random_architecture_id(self.rng)


class ReinforceSampler(Sampler):
    """A factorised categorical policy (one soft-max per decision) trained by
    REINFORCE with a moving-average baseline on the observed scores."""

    def __init__(self, space, lr=0.3, baseline_decay=0.8, seed=0):
        super(ReinforceSampler, self).__init__(space, seed)
        self.logits = [np.zeros(len(d)) for d in space.decisions()]
        self.lr, self.decay = lr, baseline_decay
        self.baseline, self.scale = None, 1.0

    def _probs(self):
        out = []
        for l in self.logits:
            e = np.exp(l - l.max())
            out.append(e / e.sum())
        return out

    def sample_next_arch(self):
        return self.space.decode([self.rng.choice(len(p), p=p)
                                  for p in self._probs()])

    def fit_sampler(self, arch_id, score):
        super(ReinforceSampler, self).fit_sampler(arch_id, score)
        if self.baseline is None:
            self.baseline = float(score)
        self.scale = max(self.scale * self.decay,
                         abs(float(score) - self.baseline), 1e-8)
        adv = (float(score) - self.baseline) / self.scale
        for l, p, c in zip(self.logits, self._probs(),
                           self.space.encode(arch_id)):
            grad = -p
            grad[c] += 1.0
            l += self.lr * adv * grad
        self.baseline = self.decay * self.baseline + \
            (1 - self.decay) * float(score)


SAMPLERS = {"sequence": SequenceSampler, "random": RandomSampler,
            "ea": RegularizedEASampler, "bohb": BohbSampler,
            "rl": ReinforceSampler}


def make(name, space, seed=0, **kw):
    return SAMPLERS[name](space, seed=seed, **kw)
