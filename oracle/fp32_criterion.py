"""Single-precision parity criterion for seabreeze_diag -- TEST INFRASTRUCTURE ONLY (checker side).

Used by bench.py's parity leg and tests/test_parity_gpu.py::test_baseline_config3_fp32_vs_oracle, so that
both hold the fp32 HIP path to the SAME rule.  The yardstick is the reference's arithmetic in double
precision on the same fp32-representable inputs (oracle/sb_oracle.f90; the reference's own fp32 build
carries ~6e-4 K of sequential window-sum noise, tests/fp32_tolerance_study.py).

What is checked, per step, on the cells of the coastal band:

* the INPUTS of the trigger -- wind speed (relative), wind direction (degrees), contrast thc (K) -- against
  fixed tolerances: what rounding to fp32 costs them;
* sb_con by ERROR PROPAGATION from those measured input errors, cell by cell
  (ref: generic/sea_breeze_diag.f90:249-259):

      sb_con = st(thc) * sw(mws),   st(t) = (|t| - 0.75) / t,   sw(m) = (11 - m) / max(1, m)
      |sb_g - sb_o| <= |sw|max * sup|st'| * (dthc + u |thc|)  +  |st|max * sup|sw'| * (dmws + 2 u 11)  +  8 u |sb_o|
      st'(t) = 0.75 / t^2,   sw'(m) = -11 / m^2 (m >= 1), -1 (m < 1)

  with dthc, dmws the differences actually found in this cell, the suprema over the interval between the
  two sides, and u = 2^-24 for the fp32 evaluation of the formula itself.  No cell is masked: a cell close
  to the 0.75 K or the 11 m/s knife edge passes exactly when its sb_con error is what its own thc / mws
  error explains.  `ratio` = |error| / bound must stay <= 1;
* trigger flips (one side zero, the other not): each must be EXPLAINED by a criterion of :249-250 whose
  margin on the fp64 side is within the measured error of its input; an unexplained flip fails.
"""
from __future__ import annotations

import numpy as np

U32 = 2.0 ** -24
THRESH_THC, THRESH_WIND, THRESH_WINDCH, THRESH_WINDDIR = 0.75, 11.0, 5.0, 90.0     # ref :127-138
TOL = {"windspeed_rel": 5e-6, "winddir_abs_deg": 1e-3, "thc_abs_K": 2e-4, "sb_con_bound_ratio": 1.0}


def _st(t):
    return (np.abs(t) - THRESH_THC) / t


def _sw(m):
    return (THRESH_WIND - m) / np.maximum(1.0, m)


def sb_con_bound(thc_g, thc_o, mws_g, mws_o, sb_o):
    """Per-cell admissible |sb_con_gpu - sb_con_oracle| given the two sides' thc and mean wind speed."""
    dthc = np.abs(thc_g - thc_o) + U32 * np.abs(thc_o)
    dmws = np.abs(mws_g - mws_o) + 2 * U32 * THRESH_WIND
    tmin = np.minimum(np.abs(thc_g), np.abs(thc_o))
    d_st = THRESH_THC / np.maximum(tmin, 1e-30) ** 2
    d_st = np.where(np.sign(thc_g) == np.sign(thc_o), d_st, np.inf)        # (a sign change is no rounding matter)
    mlo, mhi = np.minimum(mws_g, mws_o), np.maximum(mws_g, mws_o)
    d_sw = np.where(mhi < 1.0, 1.0, THRESH_WIND / np.maximum(mlo, 1.0) ** 2)
    sw_max = np.maximum(np.abs(_sw(mws_g)), np.abs(_sw(mws_o)))
    st_max = np.maximum(np.abs(_st(thc_g)), np.abs(_st(thc_o)))
    term_thc = sw_max * d_st * dthc
    term_mws = st_max * d_sw * dmws
    eps = 8 * U32 * np.abs(sb_o)
    return term_thc + term_mws + eps, term_thc, term_mws, eps


def mean_wind(tn, ws_prev, ws_new):
    """mws of ref :243 from the carried state: the state before the call and n_ws (= the state after it,
    generic flavour :261); on the first step the state is set first (:235-239)."""
    return ws_new.copy() if tn < 2 else 0.5 * (ws_prev + ws_new)


def check_step(tn, g_prev, g_new, o_prev, o_new, band, n_wd_o=None, timestep=None, target_time=6.0 * 3600.0):
    """One step.  g_*/o_*: [windspeed, winddir, thc, sb_con] before and after the call (GPU fp32 / oracle fp64),
    band: |mask| <= maxdist.  n_wd_o: this step's wind direction on the oracle side where the state does not
    show it (a step stores winddir only when tn < 2 or tn * timestep is a multiple of target_time, ref :235-239,
    :264-266); only used to explain flips.  Returns a dict of plain numbers."""
    f8 = lambda a: np.asarray(a, dtype=np.float64)
    gp, gn, op, on = [[f8(a) for a in s] for s in (g_prev, g_new, o_prev, o_new)]
    out = {}
    b = band
    den = np.maximum(np.abs(on[0][b]), 1e-2)
    out["windspeed_rel"] = float((np.abs(gn[0][b] - on[0][b]) / den).max()) if b.any() else 0.0
    dd = np.abs(gn[1][b] - on[1][b])
    out["winddir_abs_deg"] = float(np.minimum(dd, 360.0 - dd).max()) if b.any() else 0.0
    out["thc_abs_K"] = float(np.abs(gn[2][b] - on[2][b]).max()) if b.any() else 0.0
    mws_g, mws_o = mean_wind(tn, gp[0], gn[0]), mean_wind(tn, op[0], on[0])
    trig_g, trig_o = gn[3] != 0, on[3] != 0
    both = b & trig_g & trig_o
    out["triggered_cells_fp64"] = int(trig_o.sum())
    out["cells_compared"] = int(both.sum())
    out["sb_con_bound_ratio"] = 0.0
    out["sb_con_rel_max"] = 0.0
    if both.any():
        idx = np.nonzero(both)
        tg, to, mg, mo = gn[2][idx], on[2][idx], mws_g[idx], mws_o[idx]
        sg, so = gn[3][idx], on[3][idx]
        bound, t_thc, t_mws, eps = sb_con_bound(tg, to, mg, mo, so)
        err = np.abs(sg - so)
        ratio = err / bound
        k = int(np.argmax(ratio))
        out["sb_con_bound_ratio"] = float(ratio[k])
        rel = err / np.maximum(np.abs(so), 1e-2)
        kr = int(np.argmax(rel))
        out["sb_con_rel_max"] = float(rel[kr])

        def cell(kk):
            carrier = max((("thc near %.2f K" % THRESH_THC, t_thc[kk]), ("mws near %.0f m/s" % THRESH_WIND, t_mws[kk]),
                           ("fp32 evaluation", eps[kk])), key=lambda x: x[1])[0]
            return {"lat_index": int(idx[0][kk]), "lon_index": int(idx[1][kk]), "step": int(tn),
                    "thc": [float(tg[kk]), float(to[kk])], "mws": [float(mg[kk]), float(mo[kk])],
                    "sb_con": [float(sg[kk]), float(so[kk])], "abs_err": float(err[kk]), "rel_err": float(rel[kk]),
                    "bound": float(bound[kk]), "bound_terms": {"thc": float(t_thc[kk]), "mws": float(t_mws[kk]), "eval": float(eps[kk])},
                    "carried_by": carrier, "order": "[gpu fp32, oracle fp64]"}
        out["worst_ratio_cell"] = cell(k)
        out["worst_rel_cell"] = cell(kr)
    # ---- flips: each must sit on a knife edge of :249-250 within the measured error of that criterion's input ----
    flip = b & (trig_g != trig_o)
    out["trigger_flips"] = int(flip.sum())
    out["unexplained_flips"] = 0
    if flip.any():
        idx = np.nonzero(flip)
        to, tg = on[2][idx], gn[2][idx]
        mo, mg = mws_o[idx], mws_g[idx]
        dws_o = np.zeros_like(mo) if tn < 2 else np.abs(op[0][idx] - on[0][idx])
        dws_g = np.zeros_like(mo) if tn < 2 else np.abs(gp[0][idx] - gn[0][idx])
        ok = np.abs(np.abs(to) - THRESH_THC) <= np.abs(tg - to) + U32 * np.abs(to)
        ok |= np.abs(mo - THRESH_WIND) <= np.abs(mg - mo) + 2 * U32 * THRESH_WIND
        ok |= np.abs(dws_o - THRESH_WINDCH) <= np.abs(dws_g - dws_o) + 2 * U32 * THRESH_WIND
        stored = timestep is not None and (tn * float(timestep)) % target_time < 1e-4
        if tn >= 2 and (stored or n_wd_o is not None):
            # dwd of :246-247 on the oracle side; the GPU's own differs by at most the direction error measured above
            nwd = on[1][idx] if n_wd_o is None else f8(n_wd_o)[idx]
            dwd_o = np.abs(np.mod((op[1][idx] - nwd) + 180.0, 360.0) - 180.0)
            ok |= np.abs(dwd_o - THRESH_WINDDIR) <= 2 * max(out["winddir_abs_deg"], 360.0 * U32)
        out["unexplained_flips"] = int((~ok).sum())
    return out


def merge(steps):
    """Worst case over the steps' results; `ok` against TOL plus no unexplained flip."""
    worst = {k: max(s[k] for s in steps) for k in ("windspeed_rel", "winddir_abs_deg", "thc_abs_K", "sb_con_bound_ratio", "sb_con_rel_max")}
    res = {"max_err": worst, "tolerance": dict(TOL),
           "trigger_flips": sum(s["trigger_flips"] for s in steps),
           "unexplained_flips": sum(s["unexplained_flips"] for s in steps),
           "triggered_cells_fp64": sum(s["triggered_cells_fp64"] for s in steps),
           "cells_compared": sum(s["cells_compared"] for s in steps), "masked_cells": 0}
    with_cell = [s for s in steps if "worst_ratio_cell" in s]
    if with_cell:
        res["worst_ratio_cell"] = max(with_cell, key=lambda s: s["sb_con_bound_ratio"])["worst_ratio_cell"]
        res["worst_rel_cell"] = max(with_cell, key=lambda s: s["sb_con_rel_max"])["worst_rel_cell"]
    res["ok"] = bool(all(worst[k] <= TOL[k] for k in TOL) and res["unexplained_flips"] == 0)
    return res
