"""MATLAB/Octave-style front end: the call shapes of the reference's MEX functions
`lexlse(obj, options)` (interfaces/matlab-octave/lexlse.cpp) and `lexlsi(obj, options, active_set, x0, v0)`
(interfaces/matlab-octave/lexlsi.cpp:527-770) over the C ABI.  Indices are 0-based here (the MEX files take 1-based ones).

lexlse objectives:  {"A": (m, n), "b": (m,)}; an optional FIRST entry {"var": indices, "b": values} fixes variables
                    (lexlse.cpp:148-164).
lexlsi objectives:  {"A": (m, n), "lb": (m,), "ub": (m,)}; an optional FIRST entry {"var": indices, "lb": .., "ub": ..} holds
                    simple bounds (lexlsi.cpp:433-470).
options (both):     any field of ParametersLexLSE / ParametersLexLSI (typedefs.h:78-125, :178-294) plus, as in the MEX files,
                    "regularization_factors" (one per objective) and, for lexlse, "get_least_norm_solution" in {0, 1, 2, 3}.
"""
import numpy as np

from . import lexlsi as _lsi
from .lexlse import BatchedLexLSE

STATUS_OK = 0


def lexlse(obj, options=None, device: int = 0):
    """-> x (n,), info {"status"}, v (list of per-objective residual vectors, lexlse.cpp:221-236)"""
    opt = dict(options or {})
    objs = list(obj)
    fixed = objs.pop(0) if objs and "var" in objs[0] else None
    if not objs:
        raise ValueError("lexlse: at least one objective with A and b is needed")
    n = np.asarray(objs[0]["A"]).shape[1]
    dims = [np.asarray(o["b"]).size for o in objs]
    s = BatchedLexLSE(1, n, dims, device=device)
    s.setParameters(opt.get("tol_linear_dependence", 1e-12))
    reg_type = int(opt.get("regularization_type", 0))
    factors = opt.get("regularization_factors")
    if factors is not None:
        factors = np.asarray(factors, float)
        if factors.size == len(objs) + (fixed is not None):  # the MEX takes one per objective, the fixing one included (lexlse.cpp:166-172)
            factors = factors[(fixed is not None):]
    if reg_type or factors is not None:
        s.setRegularization(reg_type, factors, opt.get("variable_regularization_factor", 0.0))
    if fixed is not None:
        idx = np.zeros((1, n), np.uint32)
        val = np.zeros((1, n))
        k = np.asarray(fixed["var"]).size
        idx[0, :k] = np.asarray(fixed["var"], np.uint32)
        val[0, :k] = np.asarray(fixed["b"], float)
        s.fixVariables(np.array([k], np.uint32), idx, val)
    lod = np.zeros((1, n + 1, sum(dims)))
    r = 0
    for o, m in zip(objs, dims):
        lod[0, :n, r:r + m] = np.asarray(o["A"], float).reshape(m, n).T
        lod[0, n, r:r + m] = np.asarray(o["b"], float)
        r += m
    s.setProblem(lod)
    s.factorize()
    ln = int(opt.get("get_least_norm_solution", 0))
    {0: s.solve, 1: s.solveLeastNorm_1, 2: s.solveLeastNorm_2, 3: s.solveLeastNorm_3}[ln]()
    x = s.get_x()[0].copy()
    v = np.split(s.get_v()[0, :sum(dims)], np.cumsum(dims)[:-1])
    return x, {"status": STATUS_OK}, v


def lexlsi(obj, options=None, active_set=None, x0=None, v0=None, device: int = 0, debug: bool = False):
    """-> x, info {status, number_of_iterations, number_of_activations, number_of_deactivations, number_of_factorizations}
    (lexlsi.cpp:640-700), v (list), active_set (list of per-objective activation flags 0..3) and — with debug=True, the MEX call with five
    outputs — d {working_set_log, active_ctr, lambda, lexqr, data, xStar [, X_mu, X_mu_rhs, residual_mu]} (lexlsi.cpp:739-770)"""
    opt = dict(options or {})
    n = None
    for o in obj:
        if "A" in o:
            n = np.asarray(o["A"]).shape[1]
            break
    if n is None:
        raise ValueError("lexlsi: at least one general objective is needed")
    factors = opt.pop("regularization_factors", None)
    solve = _lsi.lsi_solve_debug if debug else _lsi.lsi_solve
    r = solve(n, obj, active_guess=active_set, x0=x0, device=device, v0=v0, regularization_factors=factors, **opt)
    i = r["info"]
    info = {"status": i["status"], "number_of_iterations": i["iterations"], "number_of_activations": i["activations"],
            "number_of_deactivations": i["deactivations"], "number_of_factorizations": i["factorizations"]}
    if debug:
        return r["x"], info, r["v"], r["active"], r["debug"]
    return r["x"], info, r["v"], r["active"]
