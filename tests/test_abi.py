"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, exports every symbol that
include/eepacc.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import make_case, ROOT
from eepacc_mpc_casadi_matlab_amd import build as eb
from eepacc_mpc_casadi_matlab_amd import engine
from eepacc_mpc_casadi_matlab_amd._abi import SettingsHolder, SettingsPOD, Vehicle, make_vehicle


@pytest.fixture(scope="module")
def lib():
    eb.build()
    return engine.load_library()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "eepacc.h")).read()
    declared = sorted(set(re.findall(r"\b(eepacc_[a-z_0-9]+)\s*\(", hdr)))
    assert set(declared) == set(engine.ABI_SYMBOLS), set(declared) ^ set(engine.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_release_build(lib):
    """Instrumented builds (-DEEPACC_AB_TIMING, -DEEPACC_DEBUG_STATUS ...) change the meaning of the status and
    iteration outputs: tests and bench must run against a library built without extra flags."""
    assert lib.eepacc_build_flags() == b"", lib.eepacc_build_flags()
    assert eb.built_flags() == ""


def test_struct_layouts_match(lib):
    assert lib.eepacc_sizeof_settings() == C.sizeof(SettingsPOD)
    assert lib.eepacc_sizeof_vehicle() == C.sizeof(Vehicle)
    assert lib.eepacc_version() == 1


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "eepacc_mpc_casadi_matlab_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no CPU", ""), f


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a box without a GPU")
def test_fails_loudly_without_gpu(lib):
    OPT, V, *_ = make_case("ABO", 20)
    holder = SettingsHolder(OPT)
    veh = make_vehicle(V)
    h = C.c_void_p()
    rc = lib.eepacc_create(C.byref(h), C.byref(holder.pod), C.byref(veh), 0, 16)
    assert rc != 0 and not h.value
    assert lib.eepacc_last_error()
    with pytest.raises(engine.EepaccError):
        engine.Engine(OPT, V)


def test_settings_validation_messages(lib):
    """eepacc_create rejects settings the kernels do not implement before touching the GPU."""
    OPT, V, *_ = make_case("ABO", 20)
    veh = make_vehicle(V)
    for key, val, code in (("solverToUse", 2, -4), ("paramEstSetting", 3, -1), ("TVestSetting", 2, -1)):
        o = dict(OPT); o[key] = val
        holder = SettingsHolder(o)
        h = C.c_void_p()
        assert lib.eepacc_create(C.byref(h), C.byref(holder.pod), C.byref(veh), 0, 16) == code
    o = dict(OPT); o["Mb"] = [1, 0] * 10                 # the first stage cannot be blocked
    holder = SettingsHolder(o)
    h = C.c_void_p()
    assert lib.eepacc_create(C.byref(h), C.byref(holder.pod), C.byref(veh), 0, 16) == -1
    assert b"Mb[0]" in lib.eepacc_last_error()


def test_nlp_header_symbols_and_layout(lib):
    """include/eepacc_nlp.h (function evaluator of RunOpt_NLP's problem): every declared entry is exported, the
    ctypes mirror of eepacc_nlp_problem has the compiled size, and creation fails loudly without a GPU."""
    from eepacc_mpc_casadi_matlab_amd import nlp
    hdr = open(os.path.join(ROOT, "include", "eepacc_nlp.h")).read()
    declared = sorted(set(re.findall(r"\b(eepacc_(?:run_)?nlp_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == ["eepacc_nlp_car_following_start_host", "eepacc_nlp_create", "eepacc_nlp_destroy", "eepacc_nlp_eval", "eepacc_nlp_newton",
                        "eepacc_nlp_postprocess_host", "eepacc_nlp_problem_from_settings", "eepacc_nlp_riccati", "eepacc_nlp_rollout",
                        "eepacc_nlp_rowdir", "eepacc_nlp_rows", "eepacc_nlp_sizeof_problem", "eepacc_nlp_solve", "eepacc_nlp_steprule",
                        "eepacc_nlp_synchronize", "eepacc_nlp_tables_free", "eepacc_nlp_trial", "eepacc_run_nlp_host"]
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.eepacc_nlp_sizeof_problem() == C.sizeof(nlp.NlpProblemPOD)
    assert C.sizeof(nlp.NlpOptions) == 4 * 4 + 7 * 8                     # eepacc_nlp_options: four int32, seven doubles
    if _no_gpu():
        OPT, V, s_tv, v_tv = make_case("ABO", 20)
        with pytest.raises(engine.EepaccError, match="no HIP device"):
            nlp.NlpEvaluator(OPT, V)
