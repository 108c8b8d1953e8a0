"""How well a training-free score ranks architectures: rank correlations
between the scores and the accuracies the architectures reach when they are
actually trained (role of the reference's src/tools/correlation.py and
exps/ score / accuracy studies).

    kendall_tau / spearman          plain implementations (ties handled), so
                                    the tool runs where scipy is absent
    score_table(...)                scores of a set of architectures under
                                    several metrics
    ground_truth(...)               validation accuracy after a few epochs
    study(...)                      both, plus the correlation per metric
"""
import numpy as np

from . import evaluators


def _ranks(v):
    """Average ranks (1-based), ties share the mean of their positions."""
    v = np.asarray(v, np.float64)
    order = np.argsort(v, kind="stable")
    ranks = np.empty(len(v), np.float64)
    i = 0
    while i < len(v):
        j = i
        while j + 1 < len(v) and v[order[j + 1]] == v[order[i]]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def spearman(a, b):
    ra, rb = _ranks(a), _ranks(b)
    ra, rb = ra - ra.mean(), rb - rb.mean()
    den = np.sqrt((ra ** 2).sum() * (rb ** 2).sum())
    return float((ra * rb).sum() / den) if den > 0 else 0.0


def kendall_tau(a, b):
    """tau-b."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = len(a)
    conc = disc = ta = tb = 0
    for i in range(n):
        da = np.sign(a[i + 1:] - a[i])
        db = np.sign(b[i + 1:] - b[i])
        prod = da * db
        conc += int((prod > 0).sum())
        disc += int((prod < 0).sum())
        ta += int(((da == 0) & (db != 0)).sum())
        tb += int(((db == 0) & (da != 0)).sum())
    den = np.sqrt(float(conc + disc + ta) * float(conc + disc + tb))
    return float((conc - disc) / den) if den > 0 else 0.0


def score_table(space, dev, batch, arch_ids, metrics):
    """{metric: [score of each architecture]}; a metric that does not apply
    to the space's models is left out."""
    table = {}
    for name in metrics:
        ev = evaluators.P1Evaluator(space, dev, batch, metric=name)
        try:
            table[name] = [ev.p1_evaluate(a) for a in arch_ids]
        except ValueError:
            continue
    return table


def ground_truth(trainer, arch_ids, epochs):
    return [trainer.p2_evaluate(a, epochs) for a in arch_ids]


def study(space, dev, batch, trainer, arch_ids, metrics, epochs=2):
    """-> (rows, accuracies): one row per metric with its Spearman and
    Kendall correlation against the trained accuracy."""
    acc = ground_truth(trainer, arch_ids, epochs)
    rows = []
    for name, scores in score_table(space, dev, batch, arch_ids,
                                    metrics).items():
        rows.append({"metric": name, "spearman": spearman(scores, acc),
                     "kendall_tau": kendall_tau(scores, acc),
                     "scores": scores})
    rows.sort(key=lambda r: -r["spearman"])
    return rows, acc
