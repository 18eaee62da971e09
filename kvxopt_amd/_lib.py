"""ctypes binding of libkvxhip.so (the C ABI declared in include/kvxhip.h).

The library is the product: if it is missing or no HIP device is visible, numeric
entry points raise -- there is no CPU fallback anywhere in this package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KVX_LIB_PATH") or os.path.join(_HERE, "libkvxhip.so")

KVX_OK, KVX_EINVAL, KVX_ENOMEM, KVX_ENOTPOSDEF, KVX_ESYMBOLIC, KVX_ESINGULAR, KVX_EDEVICE, KVX_EPERM, KVX_ECOMM = range(9)
KVX_DIST_BCAST, KVX_DIST_ALLREDUCE, KVX_DIST_ALLREDUCE_MIN = 1, 2, 3

i64 = ctypes.c_int64
f64 = ctypes.c_double
i64p = ctypes.POINTER(ctypes.c_int64)
f64p = ctypes.POINTER(ctypes.c_double)
vp = ctypes.c_void_p


class CholOpts(ctypes.Structure):
    _fields_ = [("supernodal", ctypes.c_int32), ("ordering", ctypes.c_int32), ("postorder", ctypes.c_int32),
                ("relax_small", ctypes.c_int32), ("relax_z1", f64), ("relax_z2", f64), ("relax_z3", f64),
                ("dbound", f64), ("reserved", ctypes.c_int32 * 8)]


class DistOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("root", ctypes.c_int32), ("lo", ctypes.c_int32), ("hi", ctypes.c_int32),
                ("count", ctypes.c_int64), ("buf_dev", ctypes.c_void_p)]


DIST_COMM_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(DistOp))


def pattern_digest(*arrays):
    """128-bit digest of index arrays (dtype, length and bytes of each) for the keys of the host-side caches.  The caches are looked
    up at every call with the caller's whole pattern: hashed with xxh3 (10+ GB/s, zero-copy) where the module is there, else blake2b
    (1 GB/s: 3 ms of a 33 ms conelp call on a known structure).  Round 4: cholmod, klu and the S = G'DG plans key their caches by it too -- the bytes themselves as a key were copied and
    hashed by Python at every call: 22 ms against 1.8 for config 2's 32 MB of pattern, 125 against 13 us for ACTIVSg2000's 267 KB.)"""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(a.dtype.str.encode() + int(a.size).to_bytes(8, "little"))
        if a.size:
            h.update(memoryview(a).cast("B"))
    return h.digest()


class KktSide(ctypes.Structure):
    """kvx_kkt_side of include/kvxhip.h: one right-hand side of kvx_kkt_solve_pre_dev / _post_dev."""
    _fields_ = [("xin", vp), ("xscale", f64), ("zin", vp), ("xout", vp), ("xoscale", f64), ("zout", vp), ("zoscale", f64)]


class LpCtx(ctypes.Structure):
    """kvx_lp_ctx of include/kvxhip.h: the device pointers of one conelp run (orthant, p = 0) for the four-call iteration."""
    _fields_ = [("ml", i64), ("n", i64), ("Gp", vp), ("Gi", vp), ("Gx", vp), ("max_col", i64), ("GTp", vp), ("GTi", vp), ("GTx", vp), ("max_row", i64),
                ("plan", vp), ("F", vp), ("Sx", vp), ("x2", vp)] + [(k, vp) for k in (
                    "x", "s", "z", "c", "h", "hrx", "rx", "hrz", "rz", "lmbda", "d", "di", "ds", "dz", "dx", "x1", "z1", "th", "ws3")]


class CholInfo(ctypes.Structure):
    _fields_ = [("n", i64), ("nnz_a", i64), ("lnz", i64), ("flops", f64), ("nsuper", i64), ("lsize", i64),
                ("nlevels", i64), ("max_front", i64), ("upd_size", i64), ("is_numeric", i64), ("minor", i64),
                ("solve_rowidx", i64), ("is_ll", i64), ("dev_bytes", i64), ("lsize_local", i64), ("reserved", i64 * 2)]


_SIGS = {
    "kvx_version": (ctypes.c_char_p, []),
    "kvx_device_count": (ctypes.c_int, []),
    "kvx_current_device": (ctypes.c_int, []),
    "kvx_set_device": (ctypes.c_int, [ctypes.c_int]),
    "kvx_graph_instantiate_failures": (ctypes.c_int, []),
    "kvx_last_error": (ctypes.c_char_p, []),
    "kvx_chol_default_opts": (None, [ctypes.POINTER(CholOpts)]),
    "kvx_chol_analyze": (ctypes.c_int, [i64, i64p, i64p, ctypes.c_int, i64p, ctypes.POINTER(CholOpts), ctypes.POINTER(vp)]),
    "kvx_chol_factorize": (ctypes.c_int, [vp, f64p, i64p]),
    "kvx_chol_factorize_dev": (ctypes.c_int, [vp, vp, i64p]),
    "kvx_chol_factorize_async_dev": (ctypes.c_int, [vp, vp]),
    "kvx_chol_status": (ctypes.c_int, [vp, i64p]),
    "kvx_chol_solve": (ctypes.c_int, [vp, ctypes.c_int, f64p, i64, i64]),
    "kvx_chol_solve_dev": (ctypes.c_int, [vp, ctypes.c_int, vp, i64, i64]),
    "kvx_chol_solve_async_dev": (ctypes.c_int, [vp, ctypes.c_int, vp, i64, i64]),
    "kvx_chol_factorize_solve_dev": (ctypes.c_int, [vp, vp, vp, i64, i64, i64p]),
    "kvx_chol_factorize_solve": (ctypes.c_int, [vp, f64p, f64p, i64, i64, i64p]),
    "kvx_chol_factorize_solve_async_dev": (ctypes.c_int, [vp, vp, vp, i64, i64]),
    "kvx_chol_spsolve": (ctypes.c_int, [vp, ctypes.c_int, i64, i64p, i64p, f64p,
                                        ctypes.POINTER(i64p), ctypes.POINTER(i64p), ctypes.POINTER(f64p)]),
    "kvx_chol_diag": (ctypes.c_int, [vp, f64p]),
    "kvx_chol_get_factor": (ctypes.c_int, [vp, i64p, i64p, i64p, f64p]),
    "kvx_chol_get_info": (ctypes.c_int, [vp, ctypes.POINTER(CholInfo)]),
    "kvx_chol_get_perm": (ctypes.c_int, [vp, i64p]),
    "kvx_chol_get_supernodes": (ctypes.c_int, [vp, i64p, i64p, i64p, i64p]),
    "kvx_chol_get_front_rows": (ctypes.c_int, [vp, i64p, i64p]),
    "kvx_chol_last_timing": (ctypes.c_int, [vp, f64p, f64p]),
    "kvx_chol_last_fused_path": (ctypes.c_int, [vp]),
    "kvx_chol_prof_select": (ctypes.c_int, [vp, ctypes.c_int]),
    "kvx_chol_prof_read": (ctypes.c_int, [vp, f64p, i64p]),
    "kvx_chol_dist_map": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
                                         ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint8), f64p, f64p, f64p]),
    "kvx_chol_dist_setup": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i64p]),
    "kvx_chol_dist_groups": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int32)]),
    "kvx_chol_dist_set_xchg": (ctypes.c_int, [vp, vp, i64]),
    "kvx_chol_dist_factorize": (ctypes.c_int, [vp, vp, DIST_COMM_FN, vp, i64p]),
    "kvx_chol_dist_solve": (ctypes.c_int, [vp, vp, i64, i64, DIST_COMM_FN, vp]),
    "kvx_rccl_version": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "kvx_rccl_unique_id": (ctypes.c_int, [ctypes.c_char_p]),
    "kvx_rccl_init": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(vp)]),
    "kvx_rccl_split": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int]),
    "kvx_rccl_comm": (ctypes.c_int, [vp, ctypes.POINTER(DistOp)]),
    "kvx_rccl_allreduce_host": (ctypes.c_int, [vp, f64p, ctypes.c_int, ctypes.c_int]),
    "kvx_rccl_allgather_host": (ctypes.c_int, [vp, f64p, ctypes.c_int, f64p]),
    "kvx_rccl_barrier": (ctypes.c_int, [vp]),
    "kvx_rccl_stats": (ctypes.c_int, [vp, i64p]),
    "kvx_rccl_free": (None, [vp]),
    "kvx_chol_free": (None, [vp]),
    "kvx_free": (None, [vp]),
    "kvx_atda_plan": (ctypes.c_int, [i64, i64, i64p, i64p, i64p, i64p, ctypes.POINTER(vp)]),
    "kvx_atda_pattern": (ctypes.c_int, [vp, i64p, i64p, i64p]),
    "kvx_atda_assemble": (ctypes.c_int, [vp, f64p, f64p, f64p, f64p]),
    "kvx_atda_assemble_dev": (ctypes.c_int, [vp, vp, vp, vp, vp]),
    "kvx_atda_assemble_sq_dev": (ctypes.c_int, [vp, vp, vp, vp, vp]),
    "kvx_atda_free": (None, [vp]),
    "kvx_nt_compute_scaling_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp]),
    "kvx_nt_update_scaling_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp]),
    "kvx_nt_scale_dev": (ctypes.c_int, [i64, i64, i64, vp, vp]),
    "kvx_nt_scale2_dev": (ctypes.c_int, [i64, vp, vp, ctypes.c_int]),
    "kvx_nt_sprod_dev": (ctypes.c_int, [i64, vp, vp]),
    "kvx_nt_sinv_dev": (ctypes.c_int, [i64, vp, vp]),
    "kvx_nt_ssqr_dev": (ctypes.c_int, [i64, vp, vp]),
    "kvx_nt_sdot_dev": (ctypes.c_int, [i64, vp, vp, f64p]),
    "kvx_ntq_compute_scaling_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp, vp]),
    "kvx_ntq_update_scaling_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp, vp]),
    "kvx_ntq_scale_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, i64, i64, ctypes.c_int]),
    "kvx_ntq_scale2_dev": (ctypes.c_int, [i64, vp, vp, vp, ctypes.c_int]),
    "kvx_ntq_prod_dev": (ctypes.c_int, [i64, vp, vp, vp, ctypes.c_int]),
    "kvx_ntq_max_step_dev": (ctypes.c_int, [i64, vp, vp, vp]),
    "kvx_nts_compute_scaling_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "kvx_nts_update_scaling_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "kvx_nts_scale_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, i64, i64, ctypes.c_int, vp, i64]),
    "kvx_nts_scale2_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, ctypes.c_int]),
    "kvx_nts_prod_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, ctypes.c_int, vp]),
    "kvx_nts_dot_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp]),
    "kvx_nts_max_step_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp, vp]),
    "kvx_nts_tri_dev": (ctypes.c_int, [i64, vp, vp, vp, ctypes.c_int]),
    "kvx_nts_pack_dev": (ctypes.c_int, [i64, vp, vp, vp, vp, vp, ctypes.c_int]),
    "kvx_lp_newton_rhs_dev": (ctypes.c_int, [i64, vp, vp, f64, f64, vp, vp, vp, vp, vp]),
    "kvx_lp_step_post_dev": (ctypes.c_int, [i64, f64, vp, vp, vp, vp, vp]),
    "kvx_lp_update_dev": (ctypes.c_int, [i64, f64, vp, vp, vp, vp, vp, vp, vp]),
    "kvx_lp_update_x_dev": (ctypes.c_int, [i64, i64, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "kvx_lp_residuals_dev": (ctypes.c_int, [i64, i64, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, f64, vp, vp, vp, vp]),
    "kvx_kkt_solve_pre_dev": (ctypes.c_int, [i64, i64, vp, vp, vp, i64, vp, ctypes.c_int, vp, vp, i64]),
    "kvx_kkt_solve_post_dev": (ctypes.c_int, [i64, i64, vp, vp, vp, i64, vp, ctypes.c_int, vp, vp, i64]),
    "kvx_lp_second_half_dev": (ctypes.c_int, [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f64, f64, f64, f64p]),
    "kvx_nt_reduce_multi_dev": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int32), i64p, ctypes.POINTER(vp),
                                                ctypes.POINTER(vp), f64p]),
    "kvx_lp_iter_residuals": (ctypes.c_int, [ctypes.POINTER(LpCtx), f64, f64p]),
    "kvx_lp_iter_predictor": (ctypes.c_int, [ctypes.POINTER(LpCtx), f64, f64, f64p]),
    "kvx_lp_iter_corrector": (ctypes.c_int, [ctypes.POINTER(LpCtx), f64, f64, f64, f64, f64, f64p]),
    "kvx_lp_iter_update": (ctypes.c_int, [ctypes.POINTER(LpCtx), f64, f64, f64p]),
    "kvx_nt_max_step_dev": (ctypes.c_int, [i64, vp, f64p]),
    "kvx_vec_axpy_dev": (ctypes.c_int, [i64, f64, vp, vp]),
    "kvx_dense_gemv_dev": (ctypes.c_int, [i64, i64, i64, f64, vp, i64, vp, i64, f64, vp, i64]),
    "kvx_vec_lincomb_dev": (ctypes.c_int, [i64, f64, vp, f64, vp, vp]),
    "kvx_vec_scal_dev": (ctypes.c_int, [i64, f64, vp]),
    "kvx_vec_addc_dev": (ctypes.c_int, [i64, f64, vp]),
    "kvx_vec_fill_dev": (ctypes.c_int, [i64, f64, vp]),
    "kvx_vec_copy_dev": (ctypes.c_int, [i64, vp, vp]),
    "kvx_vec_copy_strided_dev": (ctypes.c_int, [i64, vp, i64, vp]),
    "kvx_vec_xmy_dev": (ctypes.c_int, [i64, f64, vp, vp, f64, vp]),
    "kvx_spmv_dev": (ctypes.c_int, [ctypes.c_int, i64, i64, vp, vp, vp, f64, vp, f64, vp]),
    "kvx_lu_analyze": (ctypes.c_int, [i64, i64p, i64p, f64p, ctypes.POINTER(vp)]),
    "kvx_lu_free_symbolic": (None, [vp]),
    "kvx_lu_sym_info": (ctypes.c_int, [vp, i64p]),
    "kvx_lu_sym_matching": (ctypes.c_int, [vp, i64p]),
    "kvx_lu_sym_btf": (ctypes.c_int, [vp, i64p, i64p, i64p]),
    "kvx_lu_factor": (ctypes.c_int, [vp, i64, f64p, ctypes.POINTER(vp)]),
    "kvx_lu_factor_dev": (ctypes.c_int, [vp, i64, vp, ctypes.POINTER(vp)]),
    "kvx_lu_refactor": (ctypes.c_int, [vp, i64, f64p]),
    "kvx_lu_refactor_dev": (ctypes.c_int, [vp, i64, vp]),
    "kvx_lu_free_numeric": (None, [vp]),
    "kvx_lu_num_info": (ctypes.c_int, [vp, i64p]),
    "kvx_lu_num_work": (ctypes.c_int, [vp, f64p]),
    "kvx_lu_num_graph_replays": (ctypes.c_int, [vp, i64p]),
    "kvx_lu_solve": (ctypes.c_int, [vp, ctypes.c_int, f64p, i64, i64]),
    "kvx_lu_solve_dev": (ctypes.c_int, [vp, ctypes.c_int, vp, i64, i64]),
    "kvx_lu_extract": (ctypes.c_int, [vp, i64p, ctypes.POINTER(i64p), ctypes.POINTER(i64p), ctypes.POINTER(f64p),
                                      i64p, ctypes.POINTER(i64p), ctypes.POINTER(i64p), ctypes.POINTER(f64p),
                                      i64p, ctypes.POINTER(i64p), ctypes.POINTER(i64p), ctypes.POINTER(f64p),
                                      i64p, i64p, f64p, i64p, ctypes.POINTER(i64p)]),
    "kvx_lu_det": (ctypes.c_int, [vp, f64p]),
    "kvx_spmm_t_dev": (ctypes.c_int, [i64, i64, vp, vp, vp, vp, i64, vp, i64]),
    "kvx_dense_from_ccs_dev": (ctypes.c_int, [i64, i64, vp, vp, vp, vp, i64]),
    "kvx_pack_lower_dev": (ctypes.c_int, [i64, vp, i64, vp]),
    "kvx_dev_malloc": (ctypes.c_int, [ctypes.POINTER(vp), i64]),
    "kvx_dev_free": (ctypes.c_int, [vp]),
    "kvx_dev_upload": (ctypes.c_int, [vp, vp, i64]),
    "kvx_dev_download": (ctypes.c_int, [vp, vp, i64]),
    "kvx_dev_sync": (ctypes.c_int, []),
    "kvx_dev_trim": (ctypes.c_int, []),
    "kvx_dev_mem_info": (ctypes.c_int, [i64p, i64p]),
}

_lib = None


def lib():
    """Load libkvxhip.so; raises if it was not built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: the HIP extension is not built; there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGS)


def last_error():
    return lib().kvx_last_error().decode("utf-8", "replace")


def pi(a):
    return a.ctypes.data_as(i64p)


def pd(a):
    return a.ctypes.data_as(f64p)


def as_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def require_device():
    if lib().kvx_device_count() <= 0:
        raise RuntimeError("kvxopt_amd: no HIP device visible; the numeric path has no CPU fallback")


def current_device():
    """The calling thread's current HIP device (-1 without one): part of every cache key of the host layer -- a factor, a plan
    or a KKT object lives on the device that was current when it was built."""
    return int(lib().kvx_current_device())


_CACHE_CLEARERS = []


def register_cache(clear_fn):
    """The host layer's object caches (lp._KKT_CACHE, cholmod._SYMBOLIC_CACHE, klu._LINSOLVE_CACHE) register here so that
    release_cached() -- and the retry after a MemoryError -- can give their device memory back."""
    _CACHE_CLEARERS.append(clear_fn)


def release_cached():
    """Drop every cached device object of the host layer, then hand the pool's cached blocks back to the driver."""
    for f in _CACHE_CLEARERS:
        f()
    import gc
    gc.collect()
    lib().kvx_dev_trim()


def retry_after_release(fn):
    """fn(); on MemoryError release the caches and the pool once and try again."""
    try:
        return fn()
    except MemoryError:
        release_cached()
        return fn()


def raise_for(rc, what=""):
    """Map a C-ABI status to the exception the reference raises (cholmod.c error macros)."""
    if rc == KVX_OK:
        return
    msg = last_error() or what
    if rc == KVX_EINVAL:
        raise ValueError(msg or "invalid argument")
    if rc == KVX_EPERM:
        raise ValueError(msg or "p is not a valid permutation")
    if rc == KVX_ENOMEM:
        raise MemoryError(msg)
    if rc == KVX_ESYMBOLIC:
        raise ValueError(msg or "called with symbolic factor")
    if rc == KVX_ESINGULAR:
        raise ArithmeticError(msg or "singular matrix")
    if rc == KVX_ENOTPOSDEF:
        raise ArithmeticError(msg)
    raise RuntimeError("kvxhip device error: %s" % (msg or rc))


class DeviceBuffer:
    """A plain HBM allocation owned through the C ABI (no torch dependency)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = vp()
        raise_for(lib().kvx_dev_malloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        b.upload(a)
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        raise_for(lib().kvx_dev_upload(self.ptr, a.ctypes.data, a.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        raise_for(lib().kvx_dev_download(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if getattr(self, "ptr", None):
            lib().kvx_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
