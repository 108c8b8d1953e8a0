-- In-database entry points of the model-selection application, as PL/Python
-- functions over examples/model_selection/trails/pg_interface.py.  The
-- reference ships a Rust pgrx extension that embeds Python through pyo3 and
-- forwards JSON text to pg_interface.py (internal/pg_extension/src/lib.rs:
-- 34-150); function names, argument lists and the JSON-in / JSON-out
-- convention are the same here, the host language of the thin forwarding
-- layer is PL/Python instead of Rust.  Requires the plpython3u language and
-- the directory examples/model_selection on the server's PYTHONPATH.  The
-- Python functions are covered by tests/test_model_selection.py; this file
-- itself was not run (no PostgreSQL server in the build environment).
--
--   CREATE EXTENSION plpython3u;
--   \i model_selection--1.0.sql
--   SELECT model_selection(
--       (SELECT json_agg(t)::text FROM (SELECT * FROM frappe_train LIMIT 2000) t),
--       '60', '{"num_layers": 3}');

CREATE OR REPLACE FUNCTION profiling_filtering_phase(mini_batch text, config_file text)
RETURNS text AS $$
    from trails import pg_interface
    return pg_interface.profiling_filtering_phase(mini_batch, config_file)
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION profiling_refinement_phase(mini_batch text, config_file text)
RETURNS text AS $$
    from trails import pg_interface
    return pg_interface.profiling_refinement_phase(mini_batch, config_file)
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION coordinator(time_score text, time_train text, time_budget text,
                                       only_phase1 boolean, config_file text)
RETURNS text AS $$
    from trails import pg_interface
    return pg_interface.coordinator(time_score, time_train, time_budget,
                                    only_phase1, config_file)
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION filtering_phase(mini_batch text, n integer, k integer, config_file text)
RETURNS text AS $$
    from trails import pg_interface
    return pg_interface.filtering_phase(mini_batch, n, k, config_file)
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION refinement_phase(config_file text)
RETURNS text AS $$
    from trails import pg_interface
    return pg_interface.refinement_phase(config_file)
$$ LANGUAGE plpython3u;

CREATE OR REPLACE FUNCTION model_selection(mini_batch text, time_budget text, config_file text)
RETURNS text AS $$
    from trails import pg_interface
    return pg_interface.model_selection(mini_batch, time_budget, config_file)
$$ LANGUAGE plpython3u;
