"""Budget split between the two phases.  Given the measured cost of scoring
one architecture (t1) and of one training epoch (t2), a time budget T and the
ratio N / K of explored to kept architectures, choose

    N  architectures scored in the filtering phase,
    K  of them kept for refinement,
    U  epochs of the first refinement round,

such that  N * t1 + epochs(strategy, K, U) * t2 <= T  with K as large as
possible and U taken from ``u_choices`` (smallest first: more candidates
before more epochs).  Role: reference src/eva_engine/coordinator.py."""
from . import phase2


def schedule(T, t1, t2, strategy="sh", kn_rate=8, eta=3, u_choices=(1, 2, 4),
             only_phase1=False, space_size=None):
    if T <= 0:
        raise ValueError("the time budget must be positive")
    if only_phase1:
        n = int(T // max(t1, 1e-12))
        if space_size:
            n = min(n, space_size)
        return dict(K=1 if n else 0, U=0, N=n, phase1_time=n * t1,
                    phase2_time=0.0)
    best = None
    for U in u_choices:
        lo, hi = 0, 1 << 16
        while lo < hi:                      # largest feasible K by bisection
            K = (lo + hi + 1) // 2
            N = K * kn_rate
            if space_size:
                N = min(N, space_size)
            cost = N * t1 + phase2.schedule_epochs(strategy, K, U, eta) * t2
            if cost <= T and (not space_size or K <= space_size):
                lo = K
            else:
                hi = K - 1
        if lo >= 1 and (best is None or lo > best["K"]):
            N = lo * kn_rate if not space_size else min(lo * kn_rate,
                                                        space_size)
            best = dict(K=lo, U=U, N=N, phase1_time=N * t1,
                        phase2_time=phase2.schedule_epochs(strategy, lo, U,
                                                           eta) * t2)
    if best is None:
        raise ValueError("budget %.3g s is too small for one scored and one "
                         "trained architecture (t1 = %.3g, t2 = %.3g)" %
                         (T, t1, t2))
    return best
