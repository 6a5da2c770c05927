"""sine_dd on the device against known answers at 50 digits (tests/golden/sine_dd_fixtures.json, generated with mpmath by
tests/golden/make_sine_fixtures.py from the closed forms of `aggfly/aggregate/nb_kernels.py:202-251`).

Every form the planner can pick is held to the SAME exact values, so an error is attributed to a side: the reference's
libm arithmetic (restated in oracle/, see tests/test_oracle_golden.py::test_T4_oracle_against_the_50_digit_sine_fixtures:
up to ~4e-9 relative next to the window's edges, 3e-14 of the window's scale) or this engine's (table acos, rsq + Newton,
cubic arc tables — afhip_sine.h: sine_theta / sine_pair_g; afhip_kernels.h: sine_column).

Forms: the lean sine-only pair form (`_pair_ss`: BASELINE configs[4]'s kernel), the general pair form (`_pair`), the lean
four-row form (`_pair_lean_quad`), and the generic group end (6-row windows; 2- and 4-row windows with the short-group
forms switched off).  float64 cubes take every case; float32 cubes the float32-representable ones."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

# The contract (include/aggfly_hip.h): |got - exact| <= RTOL * |exact| + ATOL_SCALE * max(window range, |exact|).
RTOL, ATOL_SCALE = 1e-10, 1e-12


def _fixtures():
    fx = json.load(open(os.path.join(HERE, "golden", "sine_dd_fixtures.json")))
    return np.array(fx["ddargs"]), fx["cases"]


def _run_form(torch, hip, form, windows, dd, dtype, monkeypatch):
    """windows [L, n] -> (values [n], describe)"""
    L, n = windows.shape
    cube = torch.from_numpy(np.ascontiguousarray(windows.reshape(L, 1, n)).astype(dtype)).cuda()
    env = {"generic": {"AFHIP_NO_PAIR_MODE": "1"}}.get(form, {})
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    try:
        if form in ("lean", "generic"):        # sine_dd@group -> sum over the one group: the two-level (lean) forms
            plan = hip.FusedPlan(L, n, hip.F64 if dtype == np.float64 else hip.F32, np.array([0, L]), np.array([0, 1]),
                                 [dict(inner="sine_dd", inner_args=tuple(dd), outer="sum")])
            return plan.run_temporal(cube)[0, 0].cpu().numpy(), plan.describe()
        # single level (identity outer): what afhip_group_sine_dd builds — the general pair form for 2-row windows
        plan = hip.FusedPlan(L, n, hip.F64 if dtype == np.float64 else hip.F32, np.array([0, L]), np.array([0, 1]),
                             [dict(inner="sine_dd", inner_args=tuple(dd))])
        desc = plan.describe()
        return hip.group_sine_dd(cube, np.array([0, L]), [list(dd)]).cpu().numpy().reshape(-1).astype(np.float64), desc
    finally:
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_device_sine_forms_against_the_50_digit_fixtures(torch_cuda, monkeypatch, dtype):
    from aggfly_amd import hip
    dd_table, cases = _fixtures()
    report, seen = {}, set()
    for L, forms in ((2, ("lean", "single", "generic")), (4, ("lean", "single", "generic")), (6, ("lean",))):
        for form in forms:
            worst = {"interior": [0.0, 0.0, 0.0], "near_edge": [0.0, 0.0, 0.0]}
            n_cases, desc = 0, ""
            for row, dd in enumerate(dd_table):
                sel = [c for c in cases if c["row"] == row and len(c["window"]) == L and (dtype == np.float64 or c["f32_ok"])]
                if not sel:
                    continue
                windows = np.array([c["window"] for c in sel]).T
                got, desc = _run_form(torch_cuda, hip, form, windows, dd, dtype, monkeypatch)
                seen.add((L, form, desc.split()[0].replace("variant=", "")))
                for c, g in zip(sel, got):
                    if c["value"] is None:
                        assert np.isnan(g), (form, L, c)
                        continue
                    v, w = c["value_f64"], np.array(c["window"])
                    scale = max(w.max() - w.min(), abs(v))
                    ae = abs(float(g) - v)
                    if dtype == np.float32 and form == "single":
                        assert ae <= 1e-6 * max(abs(v), 1e-3 * scale) + 2e-7 * scale, (form, L, c, g)      # float32 OUTPUT (the reference's dtype rule)
                        continue
                    assert ae <= RTOL * abs(v) + ATOL_SCALE * scale, (form, L, desc.split()[0], c, float(g), ae)
                    t = worst["near_edge" if " * rng inside " in c["tag"] else "interior"]
                    t[0] = max(t[0], ae)
                    t[1] = max(t[1], ae / abs(v) if abs(v) > 1e-6 else 0.0)
                    t[2] = max(t[2], ae / scale)
                n_cases += len(sel)
            variants = " | ".join(sorted(v for (l2, f2, v) in seen if l2 == L and f2 == form))
            report[f"L={L} {form} ({variants})"] = {"cases": n_cases, **{k: {"max_abs": v[0], "max_rel_above_1e-6": v[1], "max_abs_over_scale": v[2]} for k, v in worst.items()}}
    # the forms really were the ones named (a threshold pair given in reverse order, t0 > t1, takes the general pair form)
    by = {}
    for L, f, v in seen:
        by.setdefault((L, f), set()).add(v)
    assert any(v.endswith("_pair_ss") for v in by[(2, "lean")]) and all("_pair" in v for v in by[(2, "lean")] | by[(2, "single")]), by
    assert any(v.endswith("_quad") for v in by[(4, "lean")]) and not any("_pair" in v for v in by[(2, "generic")] | by[(4, "generic")] | by[(4, "single")] | by[(6, "lean")]), by
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"sine_fixture_errors_{np.dtype(dtype).name}.json"), "w") as f:
            json.dump({"contract": {"rtol": RTOL, "atol_over_scale": ATOL_SCALE}, "forms": report}, f, indent=1)
