"""Parity margins: every oracle comparison of the GPU parity tests goes through `check`, which ASSERTS the bound
and RECORDS the largest deviation seen per (config, quantity).  At the end of a test session the table is written to
`gpurun_out/parity_margins.json` (tests/conftest.py); `tests/probes/margins_md.py` turns it into `profiles/rNN_parity_margins.md`.

The contract is SURVEY.md 8(d) "Parity tolerances (fp64)":

    W          max |dW| <= 1e-11 * max|W|
    logdet     |d logdet| <= 1e-10            (recorded both absolutely and relative to 1 + |logdet|)
    mu         max |d mu| <= 1e-10 * (1 + max|mu|)
    draws      <= 1e-10 relative per column (identical u)
    logq, logp <= 1e-9 * (1 + |.|)
    ELBO, SE   <= 1e-10 relative
    PSIS       log-weights / weights <= 1e-10 relative, pareto_k <= 1e-8
    indices    bit-exact

`dev` is always the already NORMALISED deviation (a pure number to compare with `bound`).  Where a test has to assert a bound looser
than the contract's, it passes `contract=` (the contract's value) and `why=` (the measured reason); both land in the table.
"""
import json
import os

import numpy as np

CONTRACT = {"W": 1e-11, "logdet_abs": 1e-10, "logdet": 1e-10, "mu": 1e-10, "draws": 1e-10, "logq": 1e-9, "logp": 1e-9,
            "elbo": 1e-10, "se": 1e-10, "psis_logw": 1e-10, "psis_w": 1e-10, "pareto_k": 1e-8}

_REC = {}


def rel(a, b):
    """|a - b| / (1 + |b|), elementwise (the form of the contract's `1e-x * (1 + |.|)` bounds)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / (1.0 + np.abs(b))


def record(config, quantity, dev, bound, contract=None, why=None):
    dev = np.asarray(dev, dtype=np.float64)
    n = int(dev.size)
    m = float(np.max(dev)) if n else 0.0
    base = quantity.split("@")[0]
    if contract is None:
        contract = CONTRACT.get(base)
    r = _REC.setdefault((config, quantity), {"max": 0.0, "n": 0, "bound": bound, "contract": contract, "why": why})
    r["max"] = m if (m != m) else max(r["max"], m)             # a NaN deviation sticks
    r["n"] += n
    r["bound"] = max(r["bound"], bound)
    if why and not r.get("why"):
        r["why"] = why
    return m


def check(config, quantity, dev, bound=None, contract=None, why=None, ctx=None):
    """record and assert `max(dev) <= bound` (default bound: the contract's value for `quantity`)"""
    base = quantity.split("@")[0]
    if bound is None:
        bound = CONTRACT[base]
    m = record(config, quantity, dev, bound, contract, why)
    if os.environ.get("PFMI_MARGINS_SOFT") == "1":             # survey run: record every margin, fail on nothing (tests/probes)
        return m
    assert m <= bound, (config, quantity, m, bound, ctx)
    return m


def dump(path):
    if not _REC:
        return None
    rows = [{"config": c, "quantity": q, **v} for (c, q), v in sorted(_REC.items())]
    old = []
    if os.path.exists(path):                                   # several pytest processes of one round append to the same table
        try:
            old = json.load(open(path))["rows"]
        except Exception:
            old = []
    keyed = {(r["config"], r["quantity"]): r for r in old}
    for r in rows:
        k = (r["config"], r["quantity"])
        if k in keyed:
            o = keyed[k]
            r = dict(r, max=max(o["max"], r["max"]) if r["max"] == r["max"] else r["max"], n=o["n"] + r["n"], bound=max(o["bound"], r["bound"]))
        keyed[k] = r
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"rows": [keyed[k] for k in sorted(keyed)]}, f, indent=1)
    return path
