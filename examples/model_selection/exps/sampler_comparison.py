"""Experiment: best score found after n evaluations by each sampler (random,
sequence, regularised evolution, BOHB-style density ratio, REINFORCE) -- the
reference's exps/ sampler study.

    python exps/sampler_comparison.py --evals 24
"""
from _common import finish, parser, setup

from trails import evaluators, samplers


def main(argv=None):
    ap = parser(__doc__.split("\n\n")[0])
    ap.add_argument("--evals", type=int, default=24)
    ap.add_argument("--metric", default="express_flow")
    ap.add_argument("--samplers", default="random,sequence,ea,bohb,rl")
    args = ap.parse_args(argv)
    sp, _, _, dev, batch = setup(args)
    ev = evaluators.P1Evaluator(sp, dev, batch, metric=args.metric)
    rows = []
    for name in args.samplers.split(","):
        s = samplers.make(name, sp, seed=args.seed)
        best, curve = None, []
        for _ in range(args.evals):
            arch = s.sample_next_arch()
            if arch is None:
                break
            score = ev.p1_evaluate(arch)
            s.fit_sampler(arch, score)
            best = score if best is None else max(best, score)
            curve.append(best)
        rows.append({"sampler": name, "best": best, "curve": curve})
    return finish(args, {"metric": args.metric, "rows": rows})


if __name__ == "__main__":
    main()
