"""Experiment: accuracy of the selected model as the time budget grows, for
the two-phase selection against training-only selection with the same budget
(the reference's exps/ budget-aware study).

    python exps/budget_sweep.py --budgets 5,10,20
"""
from _common import finish, parser, setup

from trails import run_ms


def _accuracies(trace, arch):
    """(epochs, accuracy) of ``arch`` in the rounds of a refinement trace
    (a round is a list of (architecture, accuracy, epochs))."""
    out = [(ep, acc) for rnd in trace for a, acc, ep in rnd if a == arch]
    return out or [(0, 0.0)]


def main(argv=None):
    ap = parser(__doc__.split("\n\n")[0])
    ap.add_argument("--budgets", default="5,10,20", help="seconds")
    ap.add_argument("--metric", default="express_flow")
    ap.add_argument("--sampler", default="ea")
    args = ap.parse_args(argv)
    sp, train, val, dev, _ = setup(args)
    rows = []
    for budget in (float(b) for b in args.budgets.split(",")):
        for only_phase1 in (False, True):
            rms = run_ms.RunModelSelection(sp, train, val, dev=dev,
                                           metric=args.metric,
                                           sampler=args.sampler,
                                           batch_size=args.batch,
                                           seed=args.seed)
            info = rms.select_model_online(budget, only_phase1=only_phase1)
            best = info["best_arch"]
            # filtering only: the pick has not been trained yet
            acc = rms.trainer.p2_evaluate(best, max(info["epochs_used"], 1)) \
                if only_phase1 or not info["trace"] else \
                max(a for _, a in _accuracies(info["trace"], best))
            rows.append({"budget_s": budget, "only_phase1": only_phase1,
                         "best": best, "accuracy": acc,
                         "N": info["plan"]["N"], "K": info["plan"]["K"],
                         "U": info["plan"]["U"],
                         "seconds": info["seconds"]})
    return finish(args, {"rows": rows})


if __name__ == "__main__":
    main()
