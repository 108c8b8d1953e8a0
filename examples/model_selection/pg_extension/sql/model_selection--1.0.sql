-- In-database entry points for the model-selection engine, as PL/Python
-- functions (the reference ships a Rust pgrx extension that embeds Python via
-- pyo3: internal/pg_extension/src/lib.rs; the UDF names and JSON-in / JSON-out
-- convention are kept).  Requires the plpython3u language and the repository's
-- examples/model_selection directory on the server's PYTHONPATH.
--
--   CREATE EXTENSION plpython3u;
--   \i model_selection--1.0.sql
--   SELECT model_selection('frappe', '{"n_candidates": 16, "top_k": 4}');

CREATE OR REPLACE FUNCTION _ms_load(tbl text, label_col text DEFAULT 'label')
RETURNS text AS $$
    import json
    rows = plpy.execute("SELECT * FROM %s" % plpy.quote_ident(tbl))
    cols = [c for c in rows.colnames() if c != label_col]
    x = [[float(r[c]) for c in cols] for r in rows]
    y = [int(r[label_col]) for r in rows]
    return json.dumps({"x": x, "y": y})
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION filtering_phase(tbl text, config text)
RETURNS text AS $$
    import json, numpy as np, engine
    d = json.loads(plpy.execute("SELECT _ms_load(%s) AS j" % plpy.quote_literal(tbl))[0]["j"])
    cfg = json.loads(config or "{}")
    x = np.asarray(d["x"], np.float32); y = np.asarray(d["y"], np.int32)
    ranked = engine.filtering_phase(engine.MlpSpace(), (x, y, int(y.max()) + 1),
                                    cfg.get("n_candidates", 16),
                                    cfg.get("sampler", "evolution"),
                                    cfg.get("score", "synflow"))
    return json.dumps([[list(a), s] for a, s in ranked])
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION refinement_phase(tbl text, candidates text, config text)
RETURNS text AS $$
    import json, numpy as np, engine
    d = json.loads(plpy.execute("SELECT _ms_load(%s) AS j" % plpy.quote_literal(tbl))[0]["j"])
    x = np.asarray(d["x"], np.float32); y = np.asarray(d["y"], np.int32)
    cut = int(0.75 * len(x))
    archs = [tuple(a) for a in json.loads(candidates)]
    arch, acc, _ = engine.refinement_phase(archs, (x[:cut], y[:cut], int(y.max()) + 1),
                                           (x[cut:], y[cut:]))
    return json.dumps({"best_arch": list(arch), "val_acc": acc})
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION model_selection(tbl text, config text)
RETURNS text AS $$
    import json, numpy as np, engine
    d = json.loads(plpy.execute("SELECT _ms_load(%s) AS j" % plpy.quote_literal(tbl))[0]["j"])
    cfg = json.loads(config or "{}")
    x = np.asarray(d["x"], np.float32); y = np.asarray(d["y"], np.int32)
    cut = int(0.75 * len(x))
    out = engine.model_selection((x[:cut], y[:cut], int(y.max()) + 1), (x[cut:], y[cut:]),
                                 n_candidates=cfg.get("n_candidates", 16),
                                 top_k=cfg.get("top_k", 4),
                                 sampler=cfg.get("sampler", "evolution"),
                                 score=cfg.get("score", "synflow"))
    return json.dumps(out, default=list)
$$ LANGUAGE plpython3u;
