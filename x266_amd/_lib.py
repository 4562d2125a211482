"""ctypes view of libx266hip.so (include/x266hip.h).  No compute lives here."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_P = ctypes.c_void_p
_SZ = ctypes.c_size_t
_U64 = ctypes.c_uint64

OP_DCT32_FWD, OP_DCT32_INV, OP_SATD8X8 = 0, 1, 2
PRESET_CLOSED_FORM, PRESET_VTM_DST7, PRESET_VTM_DCT8 = 0, 1, 2


class X266Error(RuntimeError):
    pass


def lib_path():
    return os.path.join(_HERE, "libx266hip.so")


def build_library(force=False):
    """Compile the HIP kernels + C ABI for gfx950 in-tree (hipcc cross-compiles
    without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "--no-print-directory"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return lib_path()


_lib = None


def load_library(path=None):
    """The product library (cached), or -- path given -- another build of the same ABI, bound afresh."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    default = path is None
    path = path or lib_path()
    if not os.path.exists(path):
        raise X266Error("%s is not built (run `make -C x266_amd/csrc`); "
                        "there is no Python/CPU fallback for the kernels" % os.path.basename(path))
    L = ctypes.CDLL(path)
    L.xHipVersion.restype = ctypes.c_char_p
    L.xHipLastError.restype = ctypes.c_char_p
    L.xHipLastError.argtypes = [_P]
    L.xHipCodecInit.argtypes = [ctypes.POINTER(_P), ctypes.c_int]
    L.xHipCodecFree.argtypes = [_P]
    L.xHipCodecFree.restype = None
    L.xHipDeviceCount.restype = ctypes.c_int
    L.xHipDeviceInfo.argtypes = [_P, ctypes.c_char_p, _SZ, ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_SZ)]
    L.xHipSetOption.argtypes = [_P, ctypes.c_char_p, ctypes.c_int]
    L.xHipGetOption.argtypes = [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.xHipAutotuneReport.argtypes = [_P, ctypes.c_char_p, _SZ]
    for name in ("xDct32FwdBatchDev", "xDct32InvBatchDev", "xSatd8x8BatchDev"):
        getattr(L, name).argtypes = [_P, _P, _P, _SZ, _P]
    L.xDct32FwdInvBatchDev.argtypes = [_P, _P, _P, _P, _SZ, _P]
    L.xFillResidualDev.argtypes = [_P, _P, _SZ, _U64, _U64, _P]
    L.xIntra32PredictDev.argtypes = [_P, _P, _P, _P, _P, _SZ, _P]
    L.xHipStreamCreate.argtypes = [_P, ctypes.POINTER(_P)]
    L.xHipStreamDestroy.argtypes = [_P, _P]
    L.xHipGraphBegin.argtypes = [_P, _P]
    L.xHipGraphEnd.argtypes = [_P, _P, ctypes.POINTER(_P)]
    L.xHipGraphLaunch.argtypes = [_P, _P, _P]
    L.xHipGraphFree.argtypes = [_P, _P]
    L.xHipGraphFree.restype = None
    L.xIntra32CostsDev.argtypes = [_P, _P, _P, _P, _P, _SZ, _P]
    L.xIntra32ResidualDct32Dev.argtypes = [_P, _P, _P, _P, _P, _P, _SZ, _P]
    L.xTransformFwdBatchDev.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P, _P, _SZ, _P, _P]
    L.xConvInputFmtDev.argtypes = [_P, _P, _P, _P, _P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, _P]
    L.xConvOutput420Dev.argtypes = [_P, _P, _P, ctypes.c_ssize_t, _P, _P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int, _P]
    L.xResidualLumaDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]
    L.xDct32FwdFromTilesDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P]
    L.xSatd8x8FromTilesDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P]
    L.xResidualChromaDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_size_t, _P]
    L.xDct32FwdChromaFromTilesDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_size_t, _P]
    L.xDct32FwdCtuFromTilesDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P]
    L.xSatd8x8ChromaFromTilesDev.argtypes = [_P, _P, _P, ctypes.c_int, ctypes.c_int, _P, _P, ctypes.c_size_t, _P]
    L.xSadBatchDev.argtypes = [_P, ctypes.c_int, _P, _P, _P, _SZ, _P]
    L.xTransformInvBatchDev.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P, _P, _SZ, _P, _P]
    L.xTransformTilesDev.argtypes = [_P, ctypes.c_int, _P, _P, _SZ, _P, _P, _P]
    L.xDct32PassDev.argtypes = [_P, _P, _P, _SZ, ctypes.c_int, _P]
    L.xDct32SatdFrameDev.argtypes = [_P, _P, _P, _SZ, _P, _P, _SZ, _P]
    L.xTransformSetMatrix.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P]
    L.xTransformGetMatrix.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P]
    L.xTransformUsePreset.argtypes = [_P, ctypes.c_int]
    L.xTransformPreset.argtypes = [_P]
    L.xHipMeScratchReserve.argtypes = [_P, _P, ctypes.c_int, ctypes.c_int]
    L.xSatd8x8SearchDev.argtypes = [_P, _P, ctypes.c_ssize_t, _P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, _P, _P, _P]
    L.xSad8x8SearchDev.argtypes = [_P, _P, ctypes.c_ssize_t, _P, ctypes.c_ssize_t, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, _P, _P, _P]
    for name in ("xDct32FwdBatch", "xDct32InvBatch", "xSatd8x8Batch"):
        getattr(L, name).argtypes = [_P, _P, _P, _SZ]
    L.xHipMalloc.argtypes = [_P, ctypes.POINTER(_P), _SZ]
    L.xHipFree.argtypes = [_P, _P]
    L.xHipHostAlloc.argtypes = [_P, ctypes.POINTER(_P), _SZ]
    L.xHipHostFree.argtypes = [_P, _P]
    L.xHipMemcpyH2D.argtypes = [_P, _P, _P, _SZ]
    L.xHipMemcpyD2H.argtypes = [_P, _P, _P, _SZ]
    L.xHipStreamSync.argtypes = [_P, _P]
    L.xHipTimeKernel.argtypes = [_P, ctypes.c_int, _P, _P, _SZ, ctypes.c_int, _P,
                                 ctypes.POINTER(ctypes.c_double)]
    L.xHipMemCeilingDev.argtypes = [_P, ctypes.c_int, _P, _P, _SZ, _P]
    L.xHipEventCreate.argtypes = [_P, ctypes.POINTER(_P)]
    L.xHipEventDestroy.argtypes = [_P, _P]
    L.xHipEventRecord.argtypes = [_P, _P, _P]
    L.xHipEventElapsedMs.argtypes = [_P, _P, _P, ctypes.POINTER(ctypes.c_double)]
    L.xDct32PackDiffRows.argtypes = [_P, ctypes.c_int, _P]
    L.xDct32PackDiffRows.restype = None
    L.xDct32PackDctWord.argtypes = [_P, ctypes.c_int]
    L.xDct32PackDctWord.restype = ctypes.c_uint64
    L.dct32_getDct.restype = ctypes.c_ulonglong
    L.satd8x8_getSatd.restype = ctypes.c_uint
    if default:
        _lib = L
    return L


def pack_diff_rows(mat, first_row):
    L = load_library()
    mat = np.ascontiguousarray(mat, np.int16)
    res = np.empty(32, np.uint32)
    L.xDct32PackDiffRows(_P(mat.ctypes.data), first_row, _P(res.ctypes.data))
    return res


def pack_dct_word(dct, idx):
    L = load_library()
    dct = np.ascontiguousarray(dct, np.int16)
    return int(L.xDct32PackDctWord(_P(dct.ctypes.data), idx))


class DeviceBuffer:
    """Device allocation owned through the C ABI (xHipMalloc / xHipFree)."""

    def __init__(self, codec, nbytes):
        self.codec, self.nbytes = codec, int(nbytes)
        p = _P()
        codec._check(codec.L.xHipMalloc(codec.ctx, ctypes.byref(p), self.nbytes), "xHipMalloc")
        self.ptr = p.value or 0

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.codec._check(self.codec.L.xHipMemcpyH2D(self.codec.ctx, self.ptr, arr.ctypes.data, arr.nbytes),
                          "xHipMemcpyH2D")

    def download(self, dtype, count):
        out = np.empty(count, dtype)
        assert out.nbytes <= self.nbytes
        self.codec._check(self.codec.L.xHipMemcpyD2H(self.codec.ctx, out.ctypes.data, self.ptr, out.nbytes),
                          "xHipMemcpyD2H")
        return out

    def free(self):
        if self.ptr:
            self.codec.L.xHipFree(self.codec.ctx, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Codec:
    """Context object mirroring x266hip_ctx (xHipCodecInit / xHipCodecFree)."""

    def __init__(self, device=0, library=None):
        """library: another build of the same ABI (tests: libx266hip_waits0.so) instead of the product x266_amd/libx266hip.so"""
        self.L = load_library(library)
        ctx = _P()
        rc = self.L.xHipCodecInit(ctypes.byref(ctx), int(device))
        if rc != 0 or not ctx.value:
            raise X266Error("xHipCodecInit(device=%d) failed with %d: no gfx950 device; "
                            "libx266hip has no CPU path" % (device, rc))
        self.ctx = ctx

    @classmethod
    def borrowed(cls, ctx_ptr):
        """A view of a context somebody else owns (xHipNodeCtx: a node rank's context, freed with the node)."""
        self = cls.__new__(cls)
        self.L = load_library()
        self.ctx = _P(ctx_ptr)
        self._borrowed = True
        return self

    def close(self):
        """Frees the context.  While pinned blocks of host_alloc are still referenced the context cannot go yet (xHipHostFree needs it): the
        codec is then CLOSED FOR NEW CALLS at once -- they raise -- a ResourceWarning says so, and the context itself is freed with the
        last block (`close_deferred` is True in between)."""
        with self._lock():
            ctx = getattr(self, "ctx", None) or getattr(self, "_ctx_for_free", None)
            if not ctx:
                return
            if getattr(self, "_pinned_live", 0) > 0:
                if getattr(self, "ctx", None):
                    import warnings
                    warnings.warn("Codec.close(): %d pinned host block(s) still referenced; the codec refuses new calls now, its context is "
                                  "freed with the last block" % self._pinned_live, ResourceWarning, stacklevel=2)
                self._ctx_for_free, self.ctx = ctx, None
                return
            if not getattr(self, "_borrowed", False):
                self.L.xHipCodecFree(ctx)
            self.ctx = self._ctx_for_free = None

    @property
    def close_deferred(self):
        return bool(getattr(self, "_ctx_for_free", None))

    def _lock(self):
        lk = self.__dict__.get("_lk")
        if lk is None:
            import threading
            lk = self.__dict__.setdefault("_lk", threading.RLock())
        return lk

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            if not getattr(self, "ctx", None):
                raise X266Error("%s: the codec is closed" % what)
            raise X266Error("%s failed (%d): %s" % (what, rc, self.L.xHipLastError(self.ctx).decode()))

    # -- info / options ----------------------------------------------------
    def device_info(self):
        name = ctypes.create_string_buffer(256)
        cu, mhz, mem = ctypes.c_int(), ctypes.c_int(), _SZ()
        self._check(self.L.xHipDeviceInfo(self.ctx, name, 256, ctypes.byref(cu), ctypes.byref(mhz),
                                          ctypes.byref(mem)), "xHipDeviceInfo")
        return {"name": name.value.decode(), "cu_count": cu.value, "clock_mhz": mhz.value,
                "hbm_bytes": mem.value}

    def set_option(self, key, value):
        self._check(self.L.xHipSetOption(self.ctx, key.encode(), int(value)), "xHipSetOption(%s)" % key)

    def get_option(self, key):
        v = ctypes.c_int()
        self._check(self.L.xHipGetOption(self.ctx, key.encode(), ctypes.byref(v)), "xHipGetOption(%s)" % key)
        return v.value

    def autotune_report(self):
        """{family: {"choice": index, "ms": [per-candidate ms]}} of the families the "autotune" option has tuned so far"""
        buf = ctypes.create_string_buffer(4096)
        self._check(self.L.xHipAutotuneReport(self.ctx, buf, len(buf)), "xHipAutotuneReport")
        out = {}
        for line in buf.value.decode().splitlines():
            w = line.split()
            out[w[0]] = {"choice": int(w[2]), "ms": [float(x) for x in w[4:]]}
        return out

    # -- host-pointer batch API (numpy in, numpy out) -------------------------
    def dct32_fwd(self, x):
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 1024)
        out = np.empty_like(x)
        self._check(self.L.xDct32FwdBatch(self.ctx, x.ctypes.data, out.ctypes.data, x.shape[0]), "xDct32FwdBatch")
        return out

    def dct32_inv(self, z):
        z = np.ascontiguousarray(z, np.int16).reshape(-1, 1024)
        out = np.empty_like(z)
        self._check(self.L.xDct32InvBatch(self.ctx, z.ctypes.data, out.ctypes.data, z.shape[0]), "xDct32InvBatch")
        return out

    def satd8x8(self, d):
        d = np.ascontiguousarray(d, np.int16).reshape(-1, 64)
        out = np.empty(d.shape[0], np.uint32)
        self._check(self.L.xSatd8x8Batch(self.ctx, d.ctypes.data, out.ctypes.data, d.shape[0]), "xSatd8x8Batch")
        return out

    # -- device-pointer batch API (raw pointers, e.g. torch .data_ptr()) --------
    def dct32_fwd_dev(self, d_in, d_out, n_blocks, stream=0):
        self._check(self.L.xDct32FwdBatchDev(self.ctx, d_in, d_out, n_blocks, stream), "xDct32FwdBatchDev")

    def dct32_inv_dev(self, d_in, d_out, n_blocks, stream=0):
        self._check(self.L.xDct32InvBatchDev(self.ctx, d_in, d_out, n_blocks, stream), "xDct32InvBatchDev")

    def dct32_fwd_inv_dev(self, d_in, d_coef, d_recon, n_blocks, stream=0):
        self._check(self.L.xDct32FwdInvBatchDev(self.ctx, d_in, d_coef or None, d_recon, n_blocks, stream), "xDct32FwdInvBatchDev")

    def intra32_predict_dev(self, d_refs, d_modes, d_ref_index, d_pred, n, stream=0):
        self._check(self.L.xIntra32PredictDev(self.ctx, d_refs, d_modes, d_ref_index or None, d_pred, n, stream), "xIntra32PredictDev")

    def intra32_predict(self, refs, modes, ref_index=None):
        """Host convenience: refs [n_refs,129] uint8 (left | top) -> predictions [n,1024] uint8."""
        refs = np.ascontiguousarray(refs, np.uint8).reshape(-1, 129)
        modes = np.ascontiguousarray(modes, np.uint8)
        n = modes.shape[0]
        padded = np.zeros((refs.shape[0], 144), np.uint8)
        padded[:, :129] = refs
        d_r, d_m, d_p = self.alloc(max(padded.nbytes, 16)), self.alloc(max(n, 16)), self.alloc(max(n * 1024, 16))
        d_r.upload(padded)
        d_m.upload(modes)
        d_i = None
        if ref_index is not None:
            d_i = self.alloc(max(4 * n, 16))
            d_i.upload(np.ascontiguousarray(ref_index, np.uint32))
        self.intra32_predict_dev(d_r.ptr, d_m.ptr, d_i.ptr if d_i else 0, d_p.ptr, n)
        self.stream_sync()
        return d_p.download(np.uint8, n * 1024).reshape(n, 1024)

    def intra32_residual_dct32_dev(self, d_refs, d_modes, d_ref_index, d_src, d_coef, n, stream=0):
        self._check(self.L.xIntra32ResidualDct32Dev(self.ctx, d_refs, d_modes, d_ref_index or None, d_src, d_coef, n, stream), "xIntra32ResidualDct32Dev")

    def intra32_residual_dct32(self, refs, modes, src, ref_index=None):
        """Host convenience: refs [n_refs,129] uint8, modes [n], src [n,1024] uint8 -> coefficients [n,1024] int16."""
        refs = np.ascontiguousarray(refs, np.uint8).reshape(-1, 129)
        modes = np.ascontiguousarray(modes, np.uint8)
        src = np.ascontiguousarray(src, np.uint8).reshape(-1, 1024)
        n = modes.shape[0]
        padded = np.zeros((refs.shape[0], 144), np.uint8)
        padded[:, :129] = refs
        d_r, d_m, d_s, d_c = self.alloc(max(padded.nbytes, 16)), self.alloc(max(n, 16)), self.alloc(max(n * 1024, 16)), self.alloc(max(n * 2048, 16))
        d_r.upload(padded)
        d_m.upload(modes)
        d_s.upload(src)
        d_i = None
        if ref_index is not None:
            d_i = self.alloc(max(4 * n, 16))
            d_i.upload(np.ascontiguousarray(ref_index, np.uint32))
        self.intra32_residual_dct32_dev(d_r.ptr, d_m.ptr, d_i.ptr if d_i else 0, d_s.ptr, d_c.ptr, n)
        self.stream_sync()
        return d_c.download(np.int16, n * 1024).reshape(n, 1024)

    def intra32_costs_dev(self, d_refs, d_src, d_costs, d_best_mode, n, stream=0):
        self._check(self.L.xIntra32CostsDev(self.ctx, d_refs, d_src, d_costs, d_best_mode or None, n, stream), "xIntra32CostsDev")

    def intra32_costs(self, refs, src):
        """Host convenience: refs [n,129], src [n,1024] uint8 -> (costs [n,35] uint32, best_mode [n] uint8)."""
        refs = np.ascontiguousarray(refs, np.uint8).reshape(-1, 129)
        src = np.ascontiguousarray(src, np.uint8).reshape(-1, 1024)
        n = refs.shape[0]
        padded = np.zeros((n, 144), np.uint8)
        padded[:, :129] = refs
        d_r, d_s = self.alloc(max(padded.nbytes, 16)), self.alloc(max(src.nbytes, 16))
        d_c, d_b = self.alloc(max(n * 35 * 4, 16)), self.alloc(max(n, 16))
        d_r.upload(padded)
        d_s.upload(src)
        self.intra32_costs_dev(d_r.ptr, d_s.ptr, d_c.ptr, d_b.ptr, n)
        self.stream_sync()
        return d_c.download(np.uint32, n * 35).reshape(n, 35), d_b.download(np.uint8, n)

    def mem_ceiling_dev(self, kind, d_src, d_dst, nbytes, stream=0):
        """kind 0 = streaming copy, 1 = read-only stream (one uint32 XOR per 2 KiB into d_dst): this box's memory ceilings"""
        self._check(self.L.xHipMemCeilingDev(self.ctx, int(kind), d_src, d_dst, nbytes, stream), "xHipMemCeilingDev")

    def satd8x8_dev(self, d_in, d_out, n_blocks, stream=0):
        self._check(self.L.xSatd8x8BatchDev(self.ctx, d_in, d_out, n_blocks, stream), "xSatd8x8BatchDev")

    def transform_fwd_dev(self, ttype, size, d_in, d_out, n_blocks, d_offsets=0, stream=0):
        self._check(self.L.xTransformFwdBatchDev(self.ctx, ttype, size, d_in, d_out, n_blocks, d_offsets or None, stream),
                    "xTransformFwdBatchDev")

    def transform_fwd(self, ttype, size, x, offsets=None, buf_samples=None):
        """numpy convenience: contiguous blocks [n, size*size], or (offsets given) a flat sample
        buffer `x` of which the blocks at `offsets` are transformed in place layout."""
        x = np.ascontiguousarray(x, np.int16)
        if offsets is None:
            n = x.size // (size * size)
            din, dout = self.alloc(max(x.nbytes, 16)), self.alloc(max(x.nbytes, 16))
            din.upload(x)
            self.transform_fwd_dev(ttype, size, din.ptr, dout.ptr, n)
            self.stream_sync()
            return dout.download(np.int16, x.size).reshape(n, size * size)
        offsets = np.ascontiguousarray(offsets, np.uint32)
        din, dout, doff = self.alloc(x.nbytes), self.alloc(x.nbytes), self.alloc(max(offsets.nbytes, 16))
        din.upload(x)
        dout.upload(np.zeros_like(x))
        doff.upload(offsets)
        self.transform_fwd_dev(ttype, size, din.ptr, dout.ptr, offsets.size, doff.ptr)
        self.stream_sync()
        return dout.download(np.int16, x.size)

    def conv_input_fmt_dev(self, d_tiles, d_y, d_u, d_v, strd_y, w, h, stream=0):
        self._check(self.L.xConvInputFmtDev(self.ctx, d_tiles, d_y, d_u, d_v, strd_y, w, h, stream), "xConvInputFmtDev")

    def conv_output_420_dev(self, d_tiles, d_y, strd_y, d_u, d_v, strd_c, w, h, stream=0):
        self._check(self.L.xConvOutput420Dev(self.ctx, d_tiles, d_y, strd_y, d_u, d_v, strd_c, w, h, stream), "xConvOutput420Dev")

    def residual_luma_dev(self, d_cur, d_pred, w, h, edge, d_res, stream=0):
        self._check(self.L.xResidualLumaDev(self.ctx, d_cur, d_pred, w, h, edge, d_res, stream), "xResidualLumaDev")

    def dct32_fwd_from_tiles_dev(self, d_cur, d_pred, w, h, d_coef, stream=0):
        self._check(self.L.xDct32FwdFromTilesDev(self.ctx, d_cur, d_pred, w, h, d_coef, stream), "xDct32FwdFromTilesDev")

    def satd8x8_from_tiles_dev(self, d_cur, d_pred, w, h, d_out, stream=0):
        self._check(self.L.xSatd8x8FromTilesDev(self.ctx, d_cur, d_pred, w, h, d_out, stream), "xSatd8x8FromTilesDev")

    def residual_chroma_dev(self, d_cur, d_pred, w, h, edge, d_res_u, d_res_v, block_pitch=1, stream=0):
        self._check(self.L.xResidualChromaDev(self.ctx, d_cur, d_pred, w, h, edge, d_res_u, d_res_v, block_pitch, stream), "xResidualChromaDev")

    def dct32_fwd_chroma_from_tiles_dev(self, d_cur, d_pred, w, h, d_coef_u, d_coef_v, block_pitch=1, stream=0):
        self._check(self.L.xDct32FwdChromaFromTilesDev(self.ctx, d_cur, d_pred, w, h, d_coef_u, d_coef_v, block_pitch, stream),
                    "xDct32FwdChromaFromTilesDev")

    def dct32_fwd_ctu_from_tiles_dev(self, d_cur, d_pred, w, h, d_coef, stream=0):
        self._check(self.L.xDct32FwdCtuFromTilesDev(self.ctx, d_cur, d_pred, w, h, d_coef, stream), "xDct32FwdCtuFromTilesDev")

    def satd8x8_chroma_from_tiles_dev(self, d_cur, d_pred, w, h, d_out_u, d_out_v, pitch=1, stream=0):
        self._check(self.L.xSatd8x8ChromaFromTilesDev(self.ctx, d_cur, d_pred, w, h, d_out_u, d_out_v, pitch, stream),
                    "xSatd8x8ChromaFromTilesDev")

    def sad_dev(self, edge, d_a, d_b, d_out, n_blocks, stream=0):
        self._check(self.L.xSadBatchDev(self.ctx, edge, d_a, d_b, d_out, n_blocks, stream), "xSadBatchDev")

    def sad(self, edge, a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, edge * edge)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, edge * edge)
        n = a.shape[0]
        da, db, do = self.alloc(max(a.nbytes, 16)), self.alloc(max(b.nbytes, 16)), self.alloc(max(4 * n, 16))
        da.upload(a)
        db.upload(b)
        self.sad_dev(edge, da.ptr, db.ptr, do.ptr, n)
        self.stream_sync()
        return do.download(np.uint32, n)

    def transform_tiles_dev(self, inverse, d_in, d_out, n_tiles, d_tile_offsets, d_tile_class, stream=0):
        self._check(self.L.xTransformTilesDev(self.ctx, int(inverse), d_in, d_out, n_tiles, d_tile_offsets or None, d_tile_class, stream),
                    "xTransformTilesDev")

    def frame_lanes_dev(self, d_dct_in, d_dct_out, n_dct, d_diff, d_satd_out, n_satd, stream=0):
        self._check(self.L.xDct32SatdFrameDev(self.ctx, d_dct_in, d_dct_out, n_dct, d_diff, d_satd_out, n_satd, stream), "xDct32SatdFrameDev")

    def dct32_pass(self, x, shift):
        """numpy convenience around xDct32PassDev: [n, 1024] int16 -> the 1-D pass of every block, stored transposed."""
        x = np.ascontiguousarray(x, np.int16).reshape(-1, 1024)
        din, dout = self.alloc(max(x.nbytes, 16)), self.alloc(max(x.nbytes, 16))
        din.upload(x)
        self._check(self.L.xDct32PassDev(self.ctx, din.ptr, dout.ptr, x.shape[0], int(shift), None), "xDct32PassDev")
        self.stream_sync()
        return dout.download(np.int16, x.size).reshape(-1, 1024)

    def set_transform_matrix(self, slot, size, m=None):
        """Install an N x N int8 matrix (row k = basis function) in 1-D transform slot 0 / 1; None restores the built-in."""
        if m is None:
            self._check(self.L.xTransformSetMatrix(self.ctx, slot, size, None), "xTransformSetMatrix")
            return
        m = np.ascontiguousarray(m, np.int8)
        assert m.shape == (size, size)
        self._check(self.L.xTransformSetMatrix(self.ctx, slot, size, m.ctypes.data), "xTransformSetMatrix")

    def use_transform_preset(self, preset):
        """0 closed-form DST-VII (built-in), 1 H.266 DST-VII as recalled from VTM, 2 DCT-VIII derived from it -- slot 1, all sizes"""
        self._check(self.L.xTransformUsePreset(self.ctx, int(preset)), "xTransformUsePreset")

    def transform_preset(self):
        return self.L.xTransformPreset(self.ctx)

    def get_transform_matrix(self, slot, size):
        m = np.empty((size, size), np.int8)
        self._check(self.L.xTransformGetMatrix(self.ctx, slot, size, m.ctypes.data), "xTransformGetMatrix")
        return m

    def transform_inv_dev(self, ttype, size, d_in, d_out, n_blocks, d_offsets=0, stream=0):
        self._check(self.L.xTransformInvBatchDev(self.ctx, ttype, size, d_in, d_out, n_blocks, d_offsets or None, stream), "xTransformInvBatchDev")

    def transform_inv(self, ttype, size, x):
        x = np.ascontiguousarray(x, np.int16)
        n = x.size // (size * size)
        din, dout = self.alloc(max(x.nbytes, 16)), self.alloc(max(x.nbytes, 16))
        din.upload(x)
        self.transform_inv_dev(ttype, size, din.ptr, dout.ptr, n)
        self.stream_sync()
        return dout.download(np.int16, x.size).reshape(n, size * size)

    def satd_search_dev(self, d_cur, cur_stride, d_ref_origin, ref_stride, width, height, rng, d_best, d_costs=0,
                        stream=0):
        self._check(self.L.xSatd8x8SearchDev(self.ctx, d_cur, cur_stride, d_ref_origin, ref_stride, width, height,
                                             rng, d_best, d_costs or None, stream), "xSatd8x8SearchDev")

    def sad_search_dev(self, d_cur, cur_stride, d_ref_origin, ref_stride, width, height, rng, d_best, d_costs=0,
                       stream=0):
        self._check(self.L.xSad8x8SearchDev(self.ctx, d_cur, cur_stride, d_ref_origin, ref_stride, width, height,
                                            rng, d_best, d_costs or None, stream), "xSad8x8SearchDev")

    def satd_search(self, cur, ref_padded, pad, rng, want_costs=False, metric="satd"):
        """numpy convenience around xSatd8x8SearchDev / xSad8x8SearchDev: cur [H,W] uint8, ref_padded [H+2*pad, W+2*pad]."""
        cur = np.ascontiguousarray(cur, np.uint8)
        refp = np.ascontiguousarray(ref_padded, np.uint8)
        h, w = cur.shape
        nb = (h // 8) * (w // 8)
        ncand = (2 * rng + 1) ** 2
        dc, dr, db = self.alloc(cur.nbytes), self.alloc(refp.nbytes), self.alloc(nb * 8)
        dcost = self.alloc(nb * ncand * 4) if want_costs else None
        dc.upload(cur)
        dr.upload(refp)
        fn = self.satd_search_dev if metric == "satd" else self.sad_search_dev
        fn(dc.ptr, cur.strides[0], dr.ptr + pad * refp.strides[0] + pad, refp.strides[0], w, h, rng,
           db.ptr, dcost.ptr if want_costs else 0)
        self.stream_sync()
        raw = db.download(np.uint8, nb * 8)
        mv = raw.view(np.int16).reshape(nb, 4)[:, :2].copy()
        cost = raw.view(np.uint32).reshape(nb, 2)[:, 1].copy()
        costs = dcost.download(np.uint32, nb * ncand).reshape(nb, ncand) if want_costs else None
        return mv, cost, costs

    def fill_residual_dev(self, d_dst, n_samples, seed, first_index=0, stream=0):
        self._check(self.L.xFillResidualDev(self.ctx, d_dst, n_samples, seed, first_index, stream),
                    "xFillResidualDev")

    # -- streams and graphs (for launch-bound sequences) ---------------------------------------
    def stream_create(self):
        s = _P()
        self._check(self.L.xHipStreamCreate(self.ctx, ctypes.byref(s)), "xHipStreamCreate")
        return s.value

    def stream_destroy(self, stream):
        self._check(self.L.xHipStreamDestroy(self.ctx, stream), "xHipStreamDestroy")

    def graph_begin(self, stream):
        self._check(self.L.xHipGraphBegin(self.ctx, stream), "xHipGraphBegin")

    def graph_end(self, stream):
        g = _P()
        self._check(self.L.xHipGraphEnd(self.ctx, stream, ctypes.byref(g)), "xHipGraphEnd")
        return g.value

    def graph_launch(self, graph, stream):
        self._check(self.L.xHipGraphLaunch(self.ctx, graph, stream), "xHipGraphLaunch")

    def graph_free(self, graph):
        self.L.xHipGraphFree(self.ctx, graph)

    def stream_sync(self, stream=0):
        self._check(self.L.xHipStreamSync(self.ctx, stream), "xHipStreamSync")

    def time_kernel(self, op, d_in, d_out, n_blocks, reps, stream=0):
        ms = ctypes.c_double()
        self._check(self.L.xHipTimeKernel(self.ctx, op, d_in, d_out, n_blocks, reps, stream, ctypes.byref(ms)),
                    "xHipTimeKernel")
        return ms.value

    # -- HIP events on the launching stream (per-launch durations of any sequence of calls) -------
    def event_create(self):
        e = _P()
        self._check(self.L.xHipEventCreate(self.ctx, ctypes.byref(e)), "xHipEventCreate")
        return e.value

    def event_destroy(self, event):
        self._check(self.L.xHipEventDestroy(self.ctx, event), "xHipEventDestroy")

    def event_record(self, event, stream=0):
        self._check(self.L.xHipEventRecord(self.ctx, event, stream), "xHipEventRecord")

    def event_elapsed_ms(self, start, stop):
        ms = ctypes.c_double()
        self._check(self.L.xHipEventElapsedMs(self.ctx, start, stop, ctypes.byref(ms)), "xHipEventElapsedMs")
        return ms.value

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def host_alloc(self, shape, dtype):
        """A numpy array in page-locked host memory (xHipHostAlloc): the host-pointer batch calls move it by DMA (43 GB/s each way
        instead of 27 from pageable memory).  The allocation belongs to a buffer owner at the END of the base chain of the array
        and of every view, slice or reshape of it: it is freed when the last of them is gone, never under a live view.  A context
        closed while blocks are alive is freed with the last block."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape))
        nbytes = max(n * dtype.itemsize, 1)
        p = _P()
        self._check(self.L.xHipHostAlloc(self.ctx, ctypes.byref(p), nbytes), "xHipHostAlloc")
        owner = _PinnedBlock(self, p.value, nbytes)
        return np.asarray(owner)[: n * dtype.itemsize].view(dtype).reshape(shape)


class _PinnedBlock:
    """Owner of one pinned allocation: numpy takes the memory through __array_interface__ and keeps this object as the base of
    the array it builds, so every view's base chain ends here (a finalizer on ONE ndarray fired under live views)."""

    def __init__(self, codec, ptr, nbytes):
        self.codec, self.ptr, self.nbytes = codec, ptr, nbytes
        with codec._lock():                                             # __del__ below may run on any thread (GC)
            codec._pinned_live = getattr(codec, "_pinned_live", 0) + 1
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        codec, ptr = self.codec, self.ptr
        self.ptr = 0
        if not ptr or codec is None:
            return
        try:
            with codec._lock():
                ctx = getattr(codec, "ctx", None) or getattr(codec, "_ctx_for_free", None)
                if ctx:
                    codec.L.xHipHostFree(ctx, ptr)
                codec._pinned_live -= 1
                if codec._pinned_live == 0 and codec.close_deferred:
                    codec.close()
        except Exception:
            pass
