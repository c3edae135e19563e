"""Inequality problems through the C ABI: the reference's LexLSI active-set driver (kept on the host, C++) over the
HIP equality solver.  Problem description = list of objectives, highest priority first:
  general objective        {"A": (m x n), "lb": (m,), "ub": (m,)}          lb <= A x - v <= ub
  simple bounds (first)    {"var": (m,) 0-based indices, "lb": ..., "ub": ...}
(the reference's MATLAB front end takes the same fields, interfaces/matlab-octave/lexlsi.cpp:380-445)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

PARAM_KEYS = ["max_number_of_factorizations", "tol_linear_dependence", "tol_wrong_sign_lambda", "tol_correct_sign_lambda",
              "tol_feasibility", "cycling_handling_enabled", "cycling_max_counter", "cycling_relax_step", "deactivate_first_wrong_sign"]
PARAM_DEFAULTS = [200, 1e-12, 1e-8, 1e-12, 1e-13, 0, 50, 1e-8, 0]  # typedefs.h:268-294
# the three regularization parameters of ParametersLexLSI (typedefs.h:185-187) travel only through the *_ex entry points
REG_PARAM_KEYS = ["regularization_type", "variable_regularization_factor", "max_number_of_CG_iterations"]
REG_PARAM_DEFAULTS = [0, 0.0, 10]
INFO_KEYS = ["status", "iterations", "activations", "deactivations", "factorizations", "total_rank"]


class InfoRows(list):
    """the per-instance info records of a batch run: behaves like a list of dicts (status, iterations, ...), built from the (batch, 6)
    int32 array the library fills — lazily, on first access: a thousand small dicts are a third of a millisecond nobody has to pay who only
    looks at the solutions"""

    def __init__(self, array):
        super().__init__()
        self.array = array  # (batch, 6) int32, columns = INFO_KEYS
        self._built = False

    def _build(self):
        if not self._built:
            self._built = True
            super().extend(dict(zip(INFO_KEYS, row)) for row in self.array.tolist())

    def __len__(self):
        return self.array.shape[0]

    def __getitem__(self, i):
        self._build()
        return super().__getitem__(i)

    def __iter__(self):
        self._build()
        return super().__iter__()

    def __eq__(self, other):
        self._build()
        if isinstance(other, InfoRows):
            other._build()
        return list.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        self._build()
        return list.__repr__(self)


def pack_params(**kw) -> np.ndarray:
    vals = list(PARAM_DEFAULTS)
    for k, v in kw.items():
        vals[PARAM_KEYS.index(k)] = float(v)
    return np.array(vals, dtype=np.float64)


def pack_params_ex(**kw) -> np.ndarray:
    """the 9 parameters of pack_params followed by regularization_type, variable_regularization_factor, max_number_of_CG_iterations"""
    keys, vals = PARAM_KEYS + REG_PARAM_KEYS, list(PARAM_DEFAULTS) + list(REG_PARAM_DEFAULTS)
    for k, v in kw.items():
        vals[keys.index(k)] = float(v)
    return np.array(vals, dtype=np.float64)


def flatten(nvar: int, objectives):
    """-> dims (uint32), types (int32), data (float64, objectives back to back, column-major), var_index (uint32)"""
    dims, types, chunks, var_index = [], [], [], np.zeros(0, np.uint32)
    for k, o in enumerate(objectives):
        lb, ub = np.asarray(o["lb"], float), np.asarray(o["ub"], float)
        dims.append(lb.size)
        if "var" in o:
            if k != 0:
                raise ValueError("simple bounds are supported only in the first objective")
            types.append(1)
            var_index = np.asarray(o["var"], np.uint32)
            m = np.stack([lb, ub], axis=1)
        else:
            types.append(0)
            m = np.hstack([np.asarray(o["A"], float).reshape(lb.size, nvar), lb[:, None], ub[:, None]])
        chunks.append(np.asfortranarray(m).ravel(order="F"))
    data = np.concatenate(chunks) if chunks else np.zeros(0)
    return np.array(dims, np.uint32), np.array(types, np.int32), np.ascontiguousarray(data), np.ascontiguousarray(var_index)


def unflatten(nvar: int, dims, types, data, var_index=None):
    """inverse of flatten() for ONE problem: the flat column-major layout -> list of objective dicts"""
    objs, off = [], 0
    for d, t in zip(np.asarray(dims, int), np.asarray(types, int)):
        w = 2 if t == 1 else nvar + 2
        m = np.asarray(data[off:off + d * w], float).reshape(w, d).T  # column-major d x w
        off += d * w
        if t == 1:
            objs.append(dict(var=np.asarray(var_index, np.uint32)[:d].copy(), lb=m[:, 0].copy(), ub=m[:, 1].copy()))
        else:
            objs.append(dict(A=m[:, :nvar].copy(), lb=m[:, nvar].copy(), ub=m[:, nvar + 1].copy()))
    return objs


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def lsi_solve(nvar: int, objectives, active_guess=None, x0=None, device: int = 0, v0=None, regularization_factors=None, **params):
    """One LexLSI problem through the C ABI.  `v0`: per-objective initial residuals (list of arrays) or None; `regularization_factors`:
    one per objective or None; `params`: ParametersLexLSI fields (PARAM_KEYS + REG_PARAM_KEYS)."""
    dims, types, data, var_index = flatten(nvar, objectives)
    total = int(dims.sum())
    x, info = np.zeros(nvar), np.zeros(6, np.int32)
    active, v = np.zeros(total, np.uint8), np.zeros(total)
    guess = None if active_guess is None else np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint8) for g in active_guess]))
    x0a = None if x0 is None else np.ascontiguousarray(x0, np.float64)
    extended = v0 is not None or regularization_factors is not None or any(k in REG_PARAM_KEYS for k in params)
    if extended:
        v0a = None if v0 is None else np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64) for a in v0]))
        rfa = None if regularization_factors is None else np.ascontiguousarray(regularization_factors, np.float64)
        par = pack_params_ex(**params)
        capi.check(capi.lib().lexls_lsi_solve_ex(
            C.c_int(device), C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, C.c_uint32), _p(types, C.c_int32), _p(data, C.c_double),
            _p(var_index if var_index.size else None, C.c_uint32), _p(guess, C.c_uint8), _p(x0a, C.c_double), _p(v0a, C.c_double),
            _p(rfa, C.c_double), _p(par, C.c_double), C.c_uint32(len(par)), _p(x, C.c_double), _p(info, C.c_int32), _p(active, C.c_uint8),
            _p(v, C.c_double)))
    else:
        par = pack_params(**params)
        capi.check(capi.lib().lexls_lsi_solve(
            C.c_int(device), C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, C.c_uint32), _p(types, C.c_int32), _p(data, C.c_double),
            _p(var_index if var_index.size else None, C.c_uint32), _p(guess, C.c_uint8), _p(x0a, C.c_double), _p(par, C.c_double),
            _p(x, C.c_double), _p(info, C.c_int32), _p(active, C.c_uint8), _p(v, C.c_double)))
    cuts = np.cumsum(dims)[:-1]
    return dict(x=x, info=dict(zip(INFO_KEYS, info.tolist())), active=np.split(active, cuts), v=np.split(v, cuts))


class _DebugStruct(C.Structure):
    """lexls_lsi_debug of include/lexls_hip.h"""
    _fields_ = [("lambda_", C.c_void_p), ("lexqr", C.c_void_p), ("data", C.c_void_p), ("x_star", C.c_void_p), ("active_ctr", C.c_void_p),
                ("log", C.c_void_p), ("log_alpha", C.c_void_p), ("max_log", C.c_uint32), ("x_mu", C.c_void_p), ("x_mu_rhs", C.c_void_p),
                ("residual_mu", C.c_void_p), ("counts", C.c_void_p)]


def debug_buffers(nvar: int, dims, max_log: int = 4096):
    """numpy arrays for the debug outputs of one LexLSI solve (lexls_lsi_debug); shaped into the MEX structure by debug_structure()"""
    total, nobj = int(np.sum(dims)), len(dims)
    return dict(lam=np.zeros((nobj, total)), lexqr=np.zeros((nvar + 1, total)), data=np.zeros((nvar + 1, total)), x_star=np.zeros(nvar),
                active_ctr=np.zeros((total, 3), np.int32), log=np.zeros((max_log, 5), np.int32), log_alpha=np.zeros(max_log),
                x_mu=np.zeros((nobj, nvar)), x_mu_rhs=np.zeros((nobj, nvar)), residual_mu=np.zeros(total), counts=np.zeros(4, np.uint32))


def debug_structure(nvar: int, dims, buf, with_mu: bool):
    """the fields of formDebugStructure (interfaces/matlab-octave/lexlsi.cpp:77-260) from filled debug_buffers()"""
    rows, nobjl, nact, nlog = (int(c) for c in buf["counts"])
    cuts = np.cumsum(dims)[:-1]
    nlog = min(nlog, buf["log"].shape[0])
    d = {
        "working_set_log": [dict(obj_index=int(e[0]), ctr_index=int(e[1]), ctr_type=int(e[2]), alpha_or_lambda=float(a), cycling_detected=int(e[3]), rank=int(e[4]))
                            for e, a in zip(buf["log"][:nlog], buf["log_alpha"][:nlog])],
        "active_ctr": [dict(obj_index=int(e[0]), ctr_index=int(e[1]), ctr_type=int(e[2])) for e in buf["active_ctr"][:nact]],
        "lambda": np.split(np.ascontiguousarray(buf["lam"].T), cuts),  # one (dim_k x nObj) matrix per objective
        "lexqr": np.ascontiguousarray(buf["lexqr"].reshape(-1)[:rows * (nvar + 1)].reshape(nvar + 1, rows).T),
        "data": np.ascontiguousarray(buf["data"].reshape(-1)[:rows * (nvar + 1)].reshape(nvar + 1, rows).T),
        "xStar": buf["x_star"],
    }
    if with_mu:
        d.update(X_mu=np.ascontiguousarray(buf["x_mu"][:nobjl].T), X_mu_rhs=np.ascontiguousarray(buf["x_mu_rhs"][:nobjl].T), residual_mu=buf["residual_mu"][:rows])
    return d


def lsi_solve_debug(nvar: int, objectives, active_guess=None, x0=None, device: int = 0, v0=None, regularization_factors=None, max_log: int = 4096, **params):
    """lsi_solve plus the MEX front end's debug structure `d` (lexls_lsi_solve_debug): returns the lsi_solve dict with key "debug"."""
    dims, types, data, var_index = flatten(nvar, objectives)
    total = int(dims.sum())
    x, info = np.zeros(nvar), np.zeros(6, np.int32)
    active, v = np.zeros(total, np.uint8), np.zeros(total)
    guess = None if active_guess is None else np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint8) for g in active_guess]))
    x0a = None if x0 is None else np.ascontiguousarray(x0, np.float64)
    v0a = None if v0 is None else np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64) for a in v0]))
    rfa = None if regularization_factors is None else np.ascontiguousarray(regularization_factors, np.float64)
    par = pack_params_ex(**params)
    buf = debug_buffers(nvar, dims, max_log)
    ds = _DebugStruct(buf["lam"].ctypes.data, buf["lexqr"].ctypes.data, buf["data"].ctypes.data, buf["x_star"].ctypes.data, buf["active_ctr"].ctypes.data,
                      buf["log"].ctypes.data, buf["log_alpha"].ctypes.data, max_log, buf["x_mu"].ctypes.data, buf["x_mu_rhs"].ctypes.data,
                      buf["residual_mu"].ctypes.data, buf["counts"].ctypes.data)
    capi.check(capi.lib().lexls_lsi_solve_debug(
        C.c_int(device), C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, C.c_uint32), _p(types, C.c_int32), _p(data, C.c_double),
        _p(var_index if var_index.size else None, C.c_uint32), _p(guess, C.c_uint8), _p(x0a, C.c_double), _p(v0a, C.c_double),
        _p(rfa, C.c_double), _p(par, C.c_double), C.c_uint32(len(par)), _p(x, C.c_double), _p(info, C.c_int32), _p(active, C.c_uint8),
        _p(v, C.c_double), C.byref(ds)))
    cuts = np.cumsum(dims)[:-1]
    return dict(x=x, info=dict(zip(INFO_KEYS, info.tolist())), active=np.split(active, cuts), v=np.split(v, cuts),
                debug=debug_structure(nvar, dims, buf, int(params.get("regularization_type", 0)) == 7))


def lsi_solve_dat(path: str, nvar: int, one_based=True, use_active_guess=False, use_x_guess=False, device: int = 0):
    x, sol, info = np.zeros(nvar), np.zeros(nvar), np.zeros(6, np.int32)
    capi.check(capi.lib().lexls_lsi_solve_dat(C.c_int(device), path.encode(), C.c_int(one_based), C.c_int(use_active_guess), C.c_int(use_x_guess),
                                              _p(x, C.c_double), _p(info, C.c_int32), _p(sol, C.c_double)))
    return dict(x=x, solution=sol, info=dict(zip(INFO_KEYS, info.tolist())))


class PackedBatch:
    """A batch of same-structure LexLSI problems in the flat layout lexls_lsi_batch_solve takes (include/lexls_hip.h): build it once
    with pack_batch() when the same constraint data is solved repeatedly (warm starts) — flattening Python objective lists costs far
    more than the solve."""

    def __init__(self, nvar, dims, types, data, var_index):
        self.nvar, self.dims, self.types, self.data, self.var_index = nvar, dims, types, data, var_index
        self.batch, self.total = data.shape[0], int(dims.sum())


def pack_batch(nvar: int, problems) -> PackedBatch:
    flat = [flatten(nvar, objs) for objs in problems]
    dims, types = flat[0][0], flat[0][1]
    for f in flat:
        if not (np.array_equal(f[0], dims) and np.array_equal(f[1], types)):
            raise ValueError("all problems of a batch must share dims and objective types")
    data = np.ascontiguousarray(np.stack([f[2] for f in flat]))
    var_index = np.ascontiguousarray(np.stack([f[3] for f in flat])) if flat[0][3].size else None
    return PackedBatch(nvar, dims, types, data, var_index)


class LsiBatch:
    """A lock-step batch that outlives one solve (lexls_lsi_batch_create / _run / _destroy): device buffers, pinned blocks, streams and the
    host worker pool are made once for `batch` problems of one structure; every run() solves new data of that structure."""

    def __init__(self, nvar: int, dims, types, batch: int, device: int = 0):
        self.nvar, self.batch = int(nvar), int(batch)
        self.dims = np.ascontiguousarray(dims, np.uint32)
        self.types = np.ascontiguousarray(types, np.int32)
        self.total = int(self.dims.sum())
        self._h = C.c_void_p()
        capi.check(capi.lib().lexls_lsi_batch_create(C.byref(self._h), C.c_int(device), C.c_uint32(self.batch), C.c_uint32(self.nvar),
                                                     C.c_uint32(len(self.dims)), _p(self.dims, C.c_uint32), _p(self.types, C.c_int32)))

    def close(self):
        if self._h:
            capi.lib().lexls_lsi_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self) -> dict:
        """of the last run: stage counts and whether the iteration step ran on the device (LEXLS_LSI_DEVICE_STEP=1 at creation)"""
        st = np.zeros(4, np.int32)
        capi.check(capi.lib().lexls_lsi_batch_stats(self._h, _p(st, C.c_int32)))
        return dict(factorize_solve=int(st[0]), sensitivity=int(st[1]), device_step=int(st[2]), groups=int(st[3]))

    def run(self, problems, active_guess=None, x0=None, regularization_factors=None, v0=None, **params):
        """`problems`: list of objective lists or a PackedBatch of this batch's structure; other arguments as lsi_batch_solve"""
        pk = problems if isinstance(problems, PackedBatch) else pack_batch(self.nvar, problems)
        if pk.batch != self.batch or not np.array_equal(pk.dims, self.dims) or not np.array_equal(pk.types, self.types):
            raise ValueError("problems do not have the structure this batch was created for")
        batch, total, nvar = self.batch, self.total, self.nvar
        guess = None
        if isinstance(active_guess, np.ndarray):
            guess = np.ascontiguousarray(active_guess.reshape(batch, total), np.uint8)
        elif active_guess is not None:
            guess = np.ascontiguousarray(np.stack([np.concatenate([np.asarray(g, np.uint8) for g in ag]) for ag in active_guess]))
        x0a = None if x0 is None else np.ascontiguousarray(x0, np.float64)
        v0a = None if v0 is None else np.ascontiguousarray(np.asarray(v0, np.float64).reshape(batch, total))  # initial residuals, (batch, sum(dims))
        x, info = np.zeros((batch, nvar)), np.zeros((batch, 6), np.int32)
        active, v, rounds = np.zeros((batch, total), np.uint8), np.zeros((batch, total)), np.zeros(2, np.int32)
        if regularization_factors is not None or any(k in REG_PARAM_KEYS for k in params):
            par = pack_params_ex(**params)
        else:
            par = pack_params(**params)
        rfa = None if regularization_factors is None else np.ascontiguousarray(regularization_factors, np.float64)
        capi.check(capi.lib().lexls_lsi_batch_run(
            self._h, _p(pk.data, C.c_double), _p(pk.var_index, C.c_uint32), _p(guess, C.c_uint8), _p(x0a, C.c_double), _p(v0a, C.c_double), _p(rfa, C.c_double),
            _p(par, C.c_double), C.c_uint32(len(par)), _p(x, C.c_double), _p(info, C.c_int32), _p(active, C.c_uint8), _p(v, C.c_double),
            _p(rounds, C.c_int32)))
        return dict(x=x, info=InfoRows(info), active=active, v=v,
                    rounds=dict(factorize_solve=int(rounds[0]), sensitivity=int(rounds[1])), dims=self.dims)


def lsi_batch_solve(nvar: int, problems, active_guess=None, x0=None, device: int = 0, regularization_factors=None, **params):
    """Lock-step batch of LexLSI problems of one structure (BASELINE configs[4]).  `problems`: list of objective lists
    (same dims / types) or a PackedBatch; `active_guess`: per problem list of per-objective flag arrays, a (batch, total) uint8
    array, or None; `x0`: (batch, nvar) or None.  One-shot form of LsiBatch (create + run + destroy)."""
    pk = problems if isinstance(problems, PackedBatch) else pack_batch(nvar, problems)
    b = LsiBatch(nvar, pk.dims, pk.types, pk.batch, device=device)
    try:
        return b.run(pk, active_guess=active_guess, x0=x0, regularization_factors=regularization_factors, **params)
    finally:
        b.close()
