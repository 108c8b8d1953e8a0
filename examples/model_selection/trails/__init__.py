"""Two-phase, budget-aware model selection (the Trails application of the
reference, examples/model_selection/Trails/internal/ml/model_selection):

    space.py        search spaces: MLP over embedded tabular features, CNN
    samplers.py     sequence / random / regularised evolution / BOHB-style
                    density-ratio / REINFORCE architecture samplers
    evaluators.py   training-free scores on this framework's autograd
                    (synflow, gradient norm, SNIP, weight norm, NAS-WOT,
                    express-flow)
    phase2.py       successive halving, successive rejects, uniform training
                    allocation with exact epoch accounting
    coordinator.py  splits a time budget into (N explored, K kept, U epochs)
    dataset.py      libsvm-coded structured data, synthetic generator
    cache.py        score cache (sqlite) and a small HTTP score service
    run_ms.py       the end-to-end driver
    pg_interface.py JSON-in / JSON-out entry points behind the database UDFs
"""
