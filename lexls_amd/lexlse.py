"""Host-side mirror of the reference's equality solver for a BATCH of problems, over the C ABI.

``BatchedLexLSE`` keeps the reference's method names and call order
(LexLS::internal::LexLSE, include/lexls/lexlse.h: resize :67, setObjDim :1426, setParameters :1467,
fixVariables :1398, setCtrType :1548, setProblem :1511, factorize :117, solve :1015,
solveLeastNorm_1 :1052, ObjectiveSensitivity :611, get_v :1560, get_x :1587, get_lexqr :1626,
getRank :1603, getTotalRank :1503) — one object is `batch` independent problems of one capacity.

Array conventions: a problem is the reference's column-major ``cap x (nVar+1)`` LOD, passed as a
C-ordered numpy array of shape ``(batch, nVar+1, cap)``.  Every compute call goes to the HIP
library; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class BatchedLexLSE:
    def __init__(self, batch: int, nVar: int, maxObjDim, device: int = 0):
        self._h = C.c_void_p()
        self.batch, self.nVar = int(batch), int(nVar)
        self.maxObjDim = np.ascontiguousarray(maxObjDim, dtype=np.uint32)
        self.nObj = int(self.maxObjDim.size)
        self.cap = int(self.maxObjDim.sum())
        capi.check(capi.lib().lexls_lse_create(C.byref(self._h), C.c_int(device), C.c_uint32(self.batch), C.c_uint32(self.nVar),
                                               C.c_uint32(self.nObj), _ptr(self.maxObjDim, C.c_uint32)))

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if self._h:
            capi.lib().lexls_lse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int):
        capi.check(capi.lib().lexls_lse_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        capi.check(capi.lib().lexls_lse_synchronize(self._h))

    # ---- problem definition ---------------------------------------------------------------------
    def setParameters(self, tol_linear_dependence: float = 1e-12):
        capi.check(capi.lib().lexls_lse_set_tolerance(self._h, C.c_double(tol_linear_dependence)))

    def setRegularization(self, regularization_type: int, factors=None, variable_factor: float = 0.0, cg_iterations: int = 10):
        """lexlse.h:1467/:1477: LexLS::RegularizationType (0 none, 1 Tikhonov, 2 Tikhonov by CGLS, 3 R, 4 R_NO_Z, 5 RT_NO_Z, 6 RT_NO_Z by CGLS, 8 Tikhonov_2, 9 test) and
        one factor per level ((nObj,)) or per problem and level ((batch, nObj))"""
        f = None if factors is None else np.ascontiguousarray(factors, dtype=np.float64)
        per = 1 if (f is not None and f.ndim == 2) else 0
        if f is not None and f.shape not in ((self.nObj,), (self.batch, self.nObj)):
            raise ValueError("factors must be (nObj,) or (batch, nObj)")
        capi.check(capi.lib().lexls_lse_set_cg_iterations(self._h, C.c_uint32(int(cg_iterations))))
        capi.check(capi.lib().lexls_lse_set_regularization(self._h, C.c_int(int(regularization_type)), _ptr(f, C.c_double) if f is not None else None,
                                                           C.c_int(per), C.c_double(variable_factor)))

    def setObjDim(self, dims):
        dims = np.ascontiguousarray(dims, dtype=np.uint32)
        per = 1 if dims.ndim == 2 else 0
        if per and dims.shape != (self.batch, self.nObj):
            raise ValueError("dims must be (nObj,) or (batch, nObj)")
        capi.check(capi.lib().lexls_lse_set_obj_dim(self._h, _ptr(dims, C.c_uint32), C.c_int(per)))

    def fixVariables(self, nfixed, index, value, ctr_type=None):
        if nfixed is None:
            capi.check(capi.lib().lexls_lse_set_fixed(self._h, None, None, None, None))
            return
        nfixed = np.ascontiguousarray(nfixed, np.uint32)
        index = np.ascontiguousarray(index, np.uint32)
        value = np.ascontiguousarray(value, np.float64)
        assert index.shape == (self.batch, self.nVar) and value.shape == (self.batch, self.nVar)
        ctr_type = None if ctr_type is None else np.ascontiguousarray(ctr_type, np.uint8)
        capi.check(capi.lib().lexls_lse_set_fixed(self._h, _ptr(nfixed, C.c_uint32), _ptr(index, C.c_uint32), _ptr(value, C.c_double),
                                                  _ptr(ctr_type, C.c_uint8)))

    def setCtrType(self, types):
        types = np.ascontiguousarray(types, np.uint8)
        assert types.shape == (self.batch, self.cap)
        capi.check(capi.lib().lexls_lse_set_ctr_type(self._h, _ptr(types, C.c_uint8)))

    def setProblem(self, lod):
        """lod: host array (batch, nVar+1, cap); copied to the device."""
        lod = np.ascontiguousarray(lod, np.float64)
        if lod.shape != (self.batch, self.nVar + 1, self.cap):
            raise ValueError(f"lod must have shape {(self.batch, self.nVar + 1, self.cap)}, got {lod.shape}")
        capi.check(capi.lib().lexls_lse_set_problem_host(self._h, _ptr(lod, C.c_double)))

    def setProblemDevice(self, device_ptr: int):
        """Bind a device buffer (e.g. torch_tensor.data_ptr()) holding batch x cap x (nVar+1) doubles as the input."""
        capi.check(capi.lib().lexls_lse_set_problem_device(self._h, C.c_void_p(device_ptr)))

    # ---- one copy each way per round (lock-step active-set batches) -----------------------------
    _ROUND_FIELDS = ("in_bytes", "dims", "nfixed", "fixed_idx", "fixed_val", "skip", "obj_index", "row_src", "row_ld", "fixed_type", "ctr_type",
                     "out_bytes", "x", "total_rank", "found", "max_abs")

    def round_layout(self) -> dict:
        """byte offsets of the per-round arrays inside the handle's two device slabs (lexls_lse_round_layout)"""
        lay = (C.c_uint64 * len(self._ROUND_FIELDS))()
        capi.check(capi.lib().lexls_lse_round_layout(self._h, lay))
        return dict(zip(self._ROUND_FIELDS, [int(v) for v in lay]))

    def round_views(self, block: np.ndarray) -> dict:
        """typed numpy views of an in-slab-shaped uint8 block (round_layout()['in_bytes'] bytes)"""
        L, B, n, cap = self.round_layout(), self.batch, self.nVar, self.cap
        def view(off, dtype, shape):
            count = int(np.prod(shape))
            return block[off:off + count * np.dtype(dtype).itemsize].view(dtype).reshape(shape)
        return dict(dims=view(L["dims"], np.uint32, (B, self.nObj)), nfixed=view(L["nfixed"], np.uint32, (B,)),
                    fixed_idx=view(L["fixed_idx"], np.uint32, (B, n)), fixed_val=view(L["fixed_val"], np.float64, (B, n)),
                    skip=view(L["skip"], np.uint8, (B,)), obj_index=view(L["obj_index"], np.int32, (B,)),
                    row_src=view(L["row_src"], np.uint32, (B, cap)), row_ld=view(L["row_ld"], np.uint32, (B, cap)),
                    fixed_type=view(L["fixed_type"], np.uint8, (B, n)), ctr_type=view(L["ctr_type"], np.uint8, (B, cap)))

    def upload_round(self, block: np.ndarray, gather: bool = False):
        assert block.dtype == np.uint8 and block.flags.c_contiguous and block.size == self.round_layout()["in_bytes"]
        capi.check(capi.lib().lexls_lse_upload_round(self._h, block.ctypes.data_as(C.c_void_p), C.c_int(1 if gather else 0)))

    def download_round(self, with_types: bool = True) -> dict:
        L, B, n, cap = self.round_layout(), self.batch, self.nVar, self.cap
        out = np.zeros(L["out_bytes"], np.uint8)
        types = np.zeros(L["in_bytes"] - L["fixed_type"], np.uint8) if with_types else None
        capi.check(capi.lib().lexls_lse_download_round(self._h, out.ctypes.data_as(C.c_void_p), types.ctypes.data_as(C.c_void_p) if with_types else None))
        r = dict(x=out[L["x"]:L["x"] + 8 * B * n].view(np.float64).reshape(B, n), total_rank=out[L["total_rank"]:L["total_rank"] + 4 * B].view(np.uint32),
                 found=out[L["found"]:L["found"] + 12 * B].view(np.int32).reshape(B, 3), max_abs=out[L["max_abs"]:L["max_abs"] + 8 * B].view(np.float64))
        if with_types:
            o = L["ctr_type"] - L["fixed_type"]
            r["fixed_type"] = types[:B * n].reshape(B, n)
            r["ctr_type"] = types[o:o + B * cap].reshape(B, cap)
        return r

    def sensitivity_resident(self, tol_wrong_sign_lambda=1e-8, tol_correct_sign_lambda=1e-12):
        """ObjectiveSensitivity with the per-problem objective indices uploaded by upload_round"""
        capi.check(capi.lib().lexls_lse_sensitivity_resident(self._h, C.c_double(tol_wrong_sign_lambda), C.c_double(tol_correct_sign_lambda)))

    # ---- hot path -------------------------------------------------------------------------------
    def factorize(self):
        capi.check(capi.lib().lexls_lse_factorize(self._h))

    def solve(self):
        capi.check(capi.lib().lexls_lse_solve(self._h))

    def factorize_solve(self, keep_factor: bool = True):
        capi.check(capi.lib().lexls_lse_factorize_solve(self._h, C.c_int(1 if keep_factor else 0)))

    def solveLeastNorm_1(self):
        capi.check(capi.lib().lexls_lse_solve_least_norm(self._h))

    def solveLeastNorm_3(self):
        """lexlse.h:1222-1277: least-norm solution from the null-space basis of the Tikhonov family"""
        capi.check(capi.lib().lexls_lse_solve_least_norm_3(self._h))

    def solveLeastNorm_2(self):
        """lexlse.h:1138-1213: least-norm solution through the normal equations of the free variables"""
        capi.check(capi.lib().lexls_lse_solve_least_norm_2(self._h))

    def setSensitivityScan(self, on: bool):
        """while on, ObjectiveSensitivity(level) goes on through the following levels until one reports a wrong-sign multiplier or the last
        level is done (LexLSI's removal search, lexlsi.h:1121-1132, in one launch); outputs are those of the level it stopped at"""
        capi.check(capi.lib().lexls_lse_set_sensitivity_scan(self._h, C.c_int(1 if on else 0)))

    def ObjectiveSensitivity(self, ObjIndex, tol_wrong_sign_lambda=1e-8, tol_correct_sign_lambda=1e-12):
        """ObjIndex: int (all problems) or per-problem int32 array (negative = skip). Returns (found, ctr, obj, maxAbs)."""
        if np.isscalar(ObjIndex):
            capi.check(capi.lib().lexls_lse_sensitivity(self._h, None, C.c_int32(int(ObjIndex)), C.c_double(tol_wrong_sign_lambda),
                                                        C.c_double(tol_correct_sign_lambda)))
        else:
            oi = np.ascontiguousarray(ObjIndex, np.int32)
            assert oi.shape == (self.batch,)
            capi.check(capi.lib().lexls_lse_sensitivity(self._h, _ptr(oi, C.c_int32), C.c_int32(0), C.c_double(tol_wrong_sign_lambda),
                                                        C.c_double(tol_correct_sign_lambda)))
        sens = np.zeros((self.batch, 3), np.int32)
        maxabs = np.zeros(self.batch)
        capi.check(capi.lib().lexls_lse_get_sensitivity(self._h, _ptr(sens, C.c_int32), _ptr(maxabs, C.c_double)))
        return sens[:, 0].astype(bool), sens[:, 1], sens[:, 2], maxabs

    # ---- results --------------------------------------------------------------------------------
    def get_x(self):
        x = np.zeros((self.batch, self.nVar))
        capi.check(capi.lib().lexls_lse_get_x(self._h, _ptr(x, C.c_double)))
        return x

    def get_lexqr(self):
        f = np.zeros((self.batch, self.nVar + 1, self.cap))
        capi.check(capi.lib().lexls_lse_get_factor(self._h, _ptr(f, C.c_double)))
        return f

    def get_hh_scalars(self):
        h = np.zeros((self.batch, self.cap))
        capi.check(capi.lib().lexls_lse_get_hh_scalars(self._h, _ptr(h, C.c_double)))
        return h

    def get_column_permutations(self):
        p = np.zeros((self.batch, self.nVar), np.uint32)
        capi.check(capi.lib().lexls_lse_get_permutation(self._h, _ptr(p, C.c_uint32)))
        return p

    def getRanks(self):
        r = np.zeros((self.batch, self.nObj), np.uint32)
        fc = np.zeros((self.batch, self.nObj), np.uint32)
        tr = np.zeros(self.batch, np.uint32)
        capi.check(capi.lib().lexls_lse_get_ranks(self._h, _ptr(r, C.c_uint32), _ptr(fc, C.c_uint32), _ptr(tr, C.c_uint32)))
        return r, fc, tr

    def get_v(self):
        capi.check(capi.lib().lexls_lse_residual(self._h))
        v = np.zeros((self.batch, self.cap))
        capi.check(capi.lib().lexls_lse_get_v(self._h, _ptr(v, C.c_double)))
        return v

    def get_mu(self):
        """(X_mu, X_mu_rhs, residual_mu) of REGULARIZATION_TIKHONOV_1 (lexlse.h:1636-1650): (batch, nObj, nVar) twice — row k is the
        reference's column k — and (batch, cap)."""
        xm = np.zeros((self.batch, self.nObj, self.nVar))
        xr = np.zeros((self.batch, self.nObj, self.nVar))
        rm = np.zeros((self.batch, self.cap))
        capi.check(capi.lib().lexls_lse_get_mu(self._h, _ptr(xm, C.c_double), _ptr(xr, C.c_double), _ptr(rm, C.c_double)))
        return xm, xr, rm

    def getWorkspace(self):
        """[lambda_fixed; lambda] of the last ObjectiveSensitivity call, shape (batch, nVar+cap)."""
        lam = np.zeros((self.batch, self.nVar + self.cap))
        capi.check(capi.lib().lexls_lse_get_lambda(self._h, _ptr(lam, C.c_double)))
        return lam

    def getCtrType(self):
        t = np.zeros((self.batch, self.cap), np.uint8)
        capi.check(capi.lib().lexls_lse_get_ctr_type(self._h, _ptr(t, C.c_uint8)))
        return t

    def device_ptr(self, name: str) -> int:
        p = C.c_void_p()
        capi.check(capi.lib().lexls_lse_device_ptr(self._h, C.c_int(capi.ARRAY[name]), C.byref(p)))
        return int(p.value)

    def set_prefix_reuse(self, enable: bool = True):
        """lexls_lse_set_prefix_reuse: factor-keeping factorizations by the register-resident wave kernel leave what a later one needs to
        read unchanged leading levels back instead of factorizing them (the reference has no such mechanism, README.md:14)"""
        capi.check(capi.lib().lexls_lse_set_prefix_reuse(self._h, C.c_int(1 if enable else 0)))

    def prefix_reuse_ready(self) -> bool:
        return bool(capi.lib().lexls_lse_prefix_reuse_ready(self._h))

    def set_resume_levels(self, levels):
        """levels[b] = number of leading levels of problem b that are unchanged since its previous factorization (consumed by the next one)"""
        lv = np.ascontiguousarray(np.broadcast_to(np.asarray(levels, np.int32), (self.batch,)))
        capi.check(capi.lib().lexls_lse_set_resume_levels(self._h, lv.ctypes.data_as(C.c_void_p)))

    def set_kernel_policy(self, policy: int):
        """diagnostics (lexls_lse_set_kernel_policy): 0 automatic dispatch, 1 generic kernel only, 2 never the left-looking wave kernel,
        3 the left-looking wave kernel whenever the shape allows it"""
        capi.check(capi.lib().lexls_lse_set_kernel_policy(self._h, C.c_int(int(policy))))

    def last_kernel(self) -> str:
        return capi.lib().lexls_lse_last_kernel(self._h).decode()
