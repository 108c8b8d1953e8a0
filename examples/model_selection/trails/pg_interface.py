"""The functions behind the database UDFs: every one takes and returns JSON
text, carries its state between calls in this module (one database session =
one Python interpreter), and never raises into the caller -- errors come back
as ``{"Errored": message}``.

    profiling_filtering_phase(mini_batch, config)  -> {"time": t1}
    profiling_refinement_phase(mini_batch, config) -> {"time": t2}
    coordinator(t1, t2, budget, only_phase1, config) -> {"k","u","n"}
    filtering_phase(mini_batch, n, k, config)      -> {"k_models": [...]}
    refinement_phase(config)                       -> {"best_arch": ...}
    model_selection(mini_batch, budget, config)    -> all of the above

Role: reference internal/ml/model_selection/pg_interface.py and the pgrx
functions of internal/pg_extension/src/lib.rs:34-150 that call it."""
import functools
import json
import traceback

from singa_amd import device

from . import coordinator as coord
from . import dataset, run_ms, space

_session = {}


def _guard(fn):
    @functools.wraps(fn)
    def run(*a, **k):
        try:
            return json.dumps(fn(*a, **k))
        except Exception:                      # noqa: BLE001 (UDF boundary)
            return json.dumps({"Errored": traceback.format_exc()})
    return run


def _config(text):
    cfg = json.loads(text) if isinstance(text, str) and text.strip() else \
        (text or {})
    cfg.setdefault("metric", "express_flow")
    cfg.setdefault("sampler", "ea")
    cfg.setdefault("strategy", "sh")
    cfg.setdefault("batch_size", 32)
    cfg.setdefault("kn_rate", 8)
    return cfg


def _driver(mini_batch, cfg):
    data = dataset.from_mini_batch_json(mini_batch, cfg.get("nfield"),
                                        cfg.get("nfeat"))
    key = (len(data), data.nfield, data.nfeat, json.dumps(cfg, sort_keys=True))
    if _session.get("key") != key:
        train, val = data.split(0.8)
        sp = space.MlpSpace(data.nfield, data.nfeat,
                            num_layers=cfg.get("num_layers", 3),
                            layer_choices=tuple(cfg.get(
                                "layer_choices", (8, 16, 32, 64))))
        _session.clear()
        _session.update(key=key, ms=run_ms.RunModelSelection(
            sp, train, val, device.get_default_device(), cfg["metric"],
            cfg["sampler"], cfg["strategy"], batch_size=cfg["batch_size"]))
    return _session["ms"]


@_guard
def profiling_filtering_phase(mini_batch, config_file="{}"):
    return {"time": _driver(mini_batch, _config(config_file))
            .profile_filtering()}


@_guard
def profiling_refinement_phase(mini_batch, config_file="{}"):
    return {"time": _driver(mini_batch, _config(config_file))
            .profile_refinement()}


@_guard
def coordinator(time_score, time_train, time_budget, only_phase1=False,
                config_file="{}"):
    cfg = _config(config_file)
    plan = coord.schedule(float(time_budget), float(time_score),
                          float(time_train), cfg["strategy"], cfg["kn_rate"],
                          only_phase1=str(only_phase1).lower() == "true")
    return {"k": plan["K"], "u": plan["U"], "n": plan["N"]}


@_guard
def filtering_phase(mini_batch, n, k, config_file="{}"):
    ms = _driver(mini_batch, _config(config_file))
    _session["kept"] = ms.filtering_phase(int(n), int(k))
    return {"k_models": _session["kept"]}


@_guard
def refinement_phase(config_file="{}"):
    cfg = _config(config_file)
    if "ms" not in _session or "kept" not in _session:
        raise RuntimeError("run filtering_phase in this session first")
    best, epochs, _ = _session["ms"].refinement_phase(int(cfg.get("u", 1)),
                                                      _session["kept"])
    return {"best_arch": best, "epochs_used": epochs}


@_guard
def model_selection(mini_batch, time_budget, config_file="{}"):
    cfg = _config(config_file)
    out = _driver(mini_batch, cfg).select_model_online(float(time_budget),
                                                       cfg["kn_rate"])
    out.pop("trace", None)
    return out
