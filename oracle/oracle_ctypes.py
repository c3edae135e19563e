"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liblexls_oracle.so (the CPU
restatement of the reference algorithm).  Imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblexls_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_last_error.restype = C.c_char_p
        _LIB.oracle_lse_time.restype = C.c_double
    return _LIB


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def lse_run(lod, dims, nvar, maxdim=None, tol=1e-12, nfixed=None, fixed_idx=None, fixed_val=None, fixed_type=None,
            ctr_type=None, solve_option=0, sens_obj=-1, tol_wrong=1e-8, tol_correct=1e-12, nthreads=1,
            reg_type=0, reg_factors=None, var_reg=0.0, cg_iters=10):
    """lod: (batch, nVar+1, cap) C array; dims: (batch, nObj) or (nObj,). Returns dict of numpy outputs."""
    lod = np.ascontiguousarray(lod, dtype=np.float64)
    batch, ncol, cap = lod.shape
    assert ncol == nvar + 1
    dims = np.asarray(dims, dtype=np.uint32)
    if dims.ndim == 1:
        dims = np.tile(dims, (batch, 1))
    dims = np.ascontiguousarray(dims)
    nobj = dims.shape[1]
    if maxdim is None:
        maxdim = dims.max(axis=0)
        extra = cap - int(maxdim.sum())
        maxdim = maxdim.copy()
        maxdim[-1] += extra
    maxdim = np.ascontiguousarray(maxdim, dtype=np.uint32)
    assert int(maxdim.sum()) == cap
    out = dict(
        x=np.zeros((batch, nvar)), factor=np.zeros_like(lod), hh=np.zeros((batch, cap)),
        perm=np.zeros((batch, nvar), np.uint32), rank=np.zeros((batch, nobj), np.uint32),
        fcol=np.zeros((batch, nobj), np.uint32), totalrank=np.zeros(batch, np.uint32), v=np.zeros((batch, cap)),
        lam=np.zeros((batch, nvar + cap)), sens=np.zeros((batch, 3), np.int32), maxabs=np.zeros(batch),
        ctr_type_out=np.zeros((batch, cap), np.uint8))
    if nfixed is not None:
        nfixed = np.ascontiguousarray(nfixed, np.uint32)
        fixed_idx = np.ascontiguousarray(fixed_idx, np.uint32)
        fixed_val = np.ascontiguousarray(fixed_val, np.float64)
        fixed_type = None if fixed_type is None else np.ascontiguousarray(fixed_type, np.uint8)
    if ctr_type is not None:
        ctr_type = np.ascontiguousarray(ctr_type, np.uint8)
    rf = None if reg_factors is None else np.ascontiguousarray(reg_factors, np.float64)
    if int(reg_type) == 7:  # the experimental type's by-products (lexlse.h:1636-1650)
        out.update(x_mu=np.zeros((batch, nobj, nvar)), x_mu_rhs=np.zeros((batch, nobj, nvar)), residual_mu=np.zeros((batch, cap)))
        lib().oracle_lse_set_mu_outputs(_p(out["x_mu"], _dp), _p(out["x_mu_rhs"], _dp), _p(out["residual_mu"], _dp))
    lib().oracle_lse_set_regularization(C.c_int(int(reg_type)), C.c_uint32(nobj), _p(rf, _dp), C.c_double(var_reg), C.c_uint32(int(cg_iters)))
    rc = lib().oracle_lse_run(
        C.c_uint32(batch), C.c_uint32(nvar), C.c_uint32(nobj), _p(maxdim, _u32p), _p(dims, _u32p), _p(lod, _dp), C.c_double(tol),
        _p(nfixed, _u32p), _p(fixed_idx, _u32p), _p(fixed_val, _dp), _p(fixed_type, _u8p), _p(ctr_type, _u8p),
        C.c_int(solve_option), C.c_int(sens_obj), C.c_double(tol_wrong), C.c_double(tol_correct),
        _p(out["x"], _dp), _p(out["factor"], _dp), _p(out["hh"], _dp), _p(out["perm"], _u32p), _p(out["rank"], _u32p),
        _p(out["fcol"], _u32p), _p(out["totalrank"], _u32p), _p(out["v"], _dp), _p(out["lam"], _dp), _p(out["sens"], _i32p),
        _p(out["maxabs"], _dp), _p(out["ctr_type_out"], _u8p), C.c_int(nthreads))
    lib().oracle_lse_set_regularization(C.c_int(0), C.c_uint32(0), None, C.c_double(0.0), C.c_uint32(10))
    lib().oracle_lse_set_mu_outputs(None, None, None)
    if rc:
        raise RuntimeError(lib().oracle_last_error().decode())
    return out


def lse_time(lod, dims, nvar, nthreads, repeats, tol=1e-12):
    lod = np.ascontiguousarray(lod, dtype=np.float64)
    batch, ncol, cap = lod.shape
    dims = np.asarray(dims, dtype=np.uint32)
    if dims.ndim == 1:
        dims = np.tile(dims, (batch, 1))
    dims = np.ascontiguousarray(dims)
    maxdim = np.ascontiguousarray(dims.max(axis=0), dtype=np.uint32)
    x = np.zeros((batch, nvar))
    return lib().oracle_lse_time(C.c_uint32(batch), C.c_uint32(nvar), C.c_uint32(dims.shape[1]), _p(maxdim, _u32p), _p(dims, _u32p),
                                 _p(lod, _dp), C.c_double(tol), _p(x, _dp), C.c_int(nthreads), C.c_int(repeats)), x


def hardware_threads():
    return int(lib().oracle_hardware_threads())


from lexls_amd.lexlsi import flatten as flatten_lsi, pack_params, pack_params_ex, REG_PARAM_KEYS  # noqa: E402  (flat problem layout shared with the product binding)


def lsi_run(nvar, objectives, active_guess=None, x0=None, v0=None, regularization_factors=None, **params):
    dims, types, data, var_index = flatten_lsi(nvar, objectives)
    total = int(dims.sum())
    x = np.zeros(nvar)
    info = np.zeros(6, np.int32)
    active = np.zeros(total, np.uint8)
    v = np.zeros(total)
    guess = None if active_guess is None else np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint8) for g in active_guess]))
    x0a = None if x0 is None else np.ascontiguousarray(x0, np.float64)
    keys = ["status", "iterations", "activations", "deactivations", "factorizations", "total_rank"]
    if v0 is not None or regularization_factors is not None or any(k in REG_PARAM_KEYS for k in params):
        v0a = None if v0 is None else np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64) for a in v0]))
        rfa = None if regularization_factors is None else np.ascontiguousarray(regularization_factors, np.float64)
        par = pack_params_ex(**params)
        rc = lib().oracle_lsi_run_ex(C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, _u32p), _p(types, _i32p), _p(data, _dp),
                                     _p(var_index if var_index.size else None, _u32p), _p(guess, _u8p), _p(x0a, _dp), _p(v0a, _dp), _p(rfa, _dp),
                                     _p(par, _dp), _p(x, _dp), _p(info, _i32p), _p(active, _u8p), _p(v, _dp))
        if rc:
            raise RuntimeError(lib().oracle_last_error().decode())
        return dict(x=x, info=dict(zip(keys, info.tolist())), active=np.split(active, np.cumsum(dims)[:-1]), v=np.split(v, np.cumsum(dims)[:-1]))
    resumable = float(bool(params.pop("resumable", False)))
    par = np.append(pack_params(**params), resumable)
    rc = lib().oracle_lsi_run(C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, _u32p), _p(types, _i32p), _p(data, _dp),
                              _p(var_index if var_index.size else None, _u32p), _p(guess, _u8p), _p(x0a, _dp), _p(par, _dp),
                              _p(x, _dp), _p(info, _i32p), _p(active, _u8p), _p(v, _dp))
    if rc:
        raise RuntimeError(lib().oracle_last_error().decode())
    keys = ["status", "iterations", "activations", "deactivations", "factorizations", "total_rank"]
    return dict(x=x, info=dict(zip(keys, info.tolist())), active=np.split(active, np.cumsum(dims)[:-1]), v=np.split(v, np.cumsum(dims)[:-1]))


def lsi_time_batch(packed, active_guess, x0, nthreads):
    """a packed batch (lexls_amd.lexlsi.PackedBatch) of LexLSI problems through the oracle-backed driver on `nthreads` host threads, default parameters:
    (factorizations, seconds of the solves) — bench.py's CPU figure for configs[4]"""
    nf, sec = C.c_int64(0), C.c_double(0.0)
    guess = None if active_guess is None else np.ascontiguousarray(active_guess, np.uint8)
    x0a = None if x0 is None else np.ascontiguousarray(x0, np.float64)
    vi = packed.var_index if packed.var_index is not None and np.size(packed.var_index) else None
    rc = lib().oracle_lsi_time_batch(C.c_uint32(packed.batch), C.c_uint32(packed.nvar), C.c_uint32(len(packed.dims)), _p(np.ascontiguousarray(packed.dims, np.uint32), _u32p),
                                     _p(np.ascontiguousarray(packed.types, np.int32), _i32p), _p(packed.data, _dp), _p(vi, _u32p), _p(guess, _u8p), _p(x0a, _dp),
                                     C.c_int(int(nthreads)), C.byref(nf), C.byref(sec))
    if rc:
        raise RuntimeError(lib().oracle_last_error().decode())
    return int(nf.value), float(sec.value)


def lsi_run_debug(nvar, objectives, active_guess=None, x0=None, v0=None, regularization_factors=None, max_log=4096, **params):
    """lsi_run plus the debug structure of the MEX front end (oracle_lsi_run_debug): the oracle-backed twin of lexls_lsi_solve_debug"""
    from lexls_amd.lexlsi import debug_buffers, debug_structure
    dims, types, data, var_index = flatten_lsi(nvar, objectives)
    total = int(dims.sum())
    x, info, active, v = np.zeros(nvar), np.zeros(6, np.int32), np.zeros(total, np.uint8), np.zeros(total)
    guess = None if active_guess is None else np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint8) for g in active_guess]))
    x0a = None if x0 is None else np.ascontiguousarray(x0, np.float64)
    v0a = None if v0 is None else np.ascontiguousarray(np.concatenate([np.asarray(a, np.float64) for a in v0]))
    rfa = None if regularization_factors is None else np.ascontiguousarray(regularization_factors, np.float64)
    par = pack_params_ex(**params)
    b = debug_buffers(nvar, dims, max_log)
    i32p = C.POINTER(C.c_int32)
    rc = lib().oracle_lsi_run_debug(C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, _u32p), _p(types, _i32p), _p(data, _dp),
                                    _p(var_index if var_index.size else None, _u32p), _p(guess, _u8p), _p(x0a, _dp), _p(v0a, _dp), _p(rfa, _dp),
                                    _p(par, _dp), _p(x, _dp), _p(info, _i32p), _p(active, _u8p), _p(v, _dp), _p(b["lam"], _dp), _p(b["lexqr"], _dp),
                                    _p(b["data"], _dp), _p(b["x_star"], _dp), _p(b["active_ctr"], i32p), _p(b["log"], i32p), _p(b["log_alpha"], _dp),
                                    C.c_uint32(max_log), _p(b["x_mu"], _dp), _p(b["x_mu_rhs"], _dp), _p(b["residual_mu"], _dp), _p(b["counts"], _u32p))
    if rc:
        raise RuntimeError(lib().oracle_last_error().decode())
    keys = ["status", "iterations", "activations", "deactivations", "factorizations", "total_rank"]
    cuts = np.cumsum(dims)[:-1]
    return dict(x=x, info=dict(zip(keys, info.tolist())), active=np.split(active, cuts), v=np.split(v, cuts),
                debug=debug_structure(nvar, dims, b, int(params.get("regularization_type", 0)) == 7))


def lsi_lambda(nvar, objectives):
    dims, types, data, var_index = flatten_lsi(nvar, objectives)
    total = int(dims.sum())
    x = np.zeros(nvar)
    lam = np.zeros((len(dims), total))  # column-major total x nObj
    rc = lib().oracle_lsi_lambda(C.c_uint32(nvar), C.c_uint32(len(dims)), _p(dims, _u32p), _p(types, _i32p), _p(data, _dp),
                                 _p(var_index if var_index.size else None, _u32p), _p(x, _dp), _p(lam, _dp))
    if rc:
        raise RuntimeError(lib().oracle_last_error().decode())
    return x, lam.T.copy()


def lsi_run_dat(path, one_based=True, use_active_guess=False, use_x_guess=False):
    hdr = np.zeros(4, np.int32)
    if lib().oracle_dat_header(path.encode(), _p(hdr, _i32p)):
        raise RuntimeError(lib().oracle_last_error().decode())
    nvar = int(hdr[0])
    x, sol, info = np.zeros(nvar), np.zeros(nvar), np.zeros(6, np.int32)
    rc = lib().oracle_lsi_run_dat(path.encode(), C.c_int(one_based), C.c_int(use_active_guess), C.c_int(use_x_guess), _p(x, _dp), _p(info, _i32p),
                                  _p(sol, _dp))
    if rc:
        raise RuntimeError(lib().oracle_last_error().decode())
    keys = ["status", "iterations", "activations", "deactivations", "factorizations", "total_rank"]
    return dict(x=x, solution=sol if hdr[3] else None, info=dict(zip(keys, info.tolist())), header=hdr)
