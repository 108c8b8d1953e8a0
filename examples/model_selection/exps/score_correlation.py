"""Experiment: which training-free score predicts trained accuracy?  Scores a
sample of the search space under every metric, trains the same architectures
for a few epochs and prints the rank correlations (the reference's
exps/ score-versus-accuracy study).

    python exps/score_correlation.py --archs 12 --epochs 2
"""
from _common import finish, parser, setup

import numpy as np

from trails import correlation, evaluators, phase2


def main(argv=None):
    ap = parser(__doc__.split("\n\n")[0])
    ap.add_argument("--archs", type=int, default=12)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--metrics", default=",".join(sorted(evaluators.METRICS)))
    args = ap.parse_args(argv)
    sp, train, val, dev, batch = setup(args)
    rng = np.random.RandomState(args.seed)
    archs = sorted({sp.random_architecture_id(rng)
                    for _ in range(4 * args.archs)})[:args.archs]
    trainer = phase2.Trainer(sp, dev, train, val, batch_size=args.batch)
    rows, acc = correlation.study(sp, dev, batch, trainer, archs,
                                  args.metrics.split(","), args.epochs)
    return finish(args, {
        "architectures": archs, "accuracy": acc,
        "metrics": [{k: r[k] for k in ("metric", "spearman", "kendall_tau")}
                    for r in rows]})


if __name__ == "__main__":
    main()
