"""Synthetic scenario generators shared by the tests and bench.py.

S2 "closed-loop scenarios": instance i starts at (s, v) = (0, U(0,10)); its lead vehicle follows
the 2 Hz TO01_EAD speed trace circularly shifted by (13 i) mod 870 samples and scaled by
U(0.8, 1.2), starting U(6, 40) m ahead; lead distance is the forward-Euler integral of the speed
as in ABO/Run_DrivingCycle.m:37-47.  Deterministic (numpy Generator, seed 1234 + rank).
"""
from __future__ import annotations

import numpy as np


def make_s2(B: int, n_steps: int, V_TO_2Hz: np.ndarray, Ts: float = 0.5, seed: int = 1234,
            first_instance: int = 0):
    # one stream per quantity, so that instance i gets the same draw whatever the shard size
    total = first_instance + B
    v0_all = np.random.default_rng([seed, 0]).uniform(0.0, 10.0, total)
    scale_all = np.random.default_rng([seed, 1]).uniform(0.8, 1.2, total)
    gap_all = np.random.default_rng([seed, 2]).uniform(6.0, 40.0, total)
    idx = np.arange(first_instance, total)
    v0, scale, gap = v0_all[idx], scale_all[idx], gap_all[idx]
    base = np.where(np.asarray(V_TO_2Hz, dtype=np.float64) < 0.1, 0.0, V_TO_2Hz)   # Run_DrivingCycle.m:17
    L = base.size
    k = np.arange(n_steps)[:, None]
    shift = ((idx * 13) % L)[None, :]
    v_tv = base[(k + shift) % L] * scale[None, :]
    v_tv[0, :] = 0.0                                   # the reference's trace starts at standstill
    s_tv = gap[None, :] + Ts * np.cumsum(v_tv, axis=0)
    s_tv[0, :] = gap
    return dict(s0=np.zeros(B), v0=v0, a_minus1=np.zeros(B), s_tv=np.ascontiguousarray(s_tv),
                v_tv=np.ascontiguousarray(v_tv))


def make_s1(B: int, golden: dict, s_tv: np.ndarray, v_tv: np.ndarray, Ts: float = 0.5, seed: int = 1234):
    """S1 "open-loop step batch": instance i takes the golden state of step (7 i) mod 871 and
    perturbs speed by U(-0.5,0.5) m/s and gap by U(-2,2) m."""
    rng = np.random.default_rng(seed)
    n = golden["s_opt"].size
    k = (np.arange(B) * 7) % n
    dv = rng.uniform(-0.5, 0.5, B)
    dgap = rng.uniform(-2.0, 2.0, B)
    s = golden["s_opt"][k].copy()
    v = np.maximum(golden["v_opt"][k] + dv, 0.0)
    vprev = golden["v_opt"][np.maximum(k - 1, 0)]
    a_prev = np.where(k > 0, (golden["v_opt"][k] - vprev) / Ts, 0.0)
    vtv = np.where(k > 0, v_tv[k], 0.0)
    vtv_prev = np.where(k > 1, v_tv[np.maximum(k - 1, 0)], 0.0)
    return dict(s=s, v=v, a_prev=a_prev, t0=k * Ts, s_tv=s_tv[k] + dgap, v_tv=vtv,
                a_tv_prev=np.where(k > 0, (vtv - vtv_prev) / Ts, 0.0), k=k)
