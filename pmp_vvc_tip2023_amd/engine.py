"""Host-side mirror of the reference's hot-path functions, running on libpmp_hip.so.

Same names and argument meaning as the reference so callers (and the parity tests) read alike:

    reference                                           here
    Metrics.inference_pre_QBD(loader, Net_Q, Net_BD)    Engine.inference_pre_QBD(comp, qp, block_y[, block_u, block_v])
    Metrics.seq_post_process(qt, bt, dire, comp, ...)   Engine.seq_post_process(qt, bt, dire, comp, sub_numfrm, width, height, save_path)
    Inference_QBD.output_block_yuv(...)                 Engine.output_block_yuv(y, u, v, bitdepth)
    Inference_QBD.load_pretrain_model(net, path)        Engine.load_pretrain_model(net_name, qp, weights)

numpy arrays in/out for the host API; the *_device methods take raw device pointers (ints), e.g. torch
tensors' .data_ptr(), and run asynchronously on the engine's stream.  There is no CPU fallback: constructing an
Engine without a gfx950 device raises PmpError(PMP_E_NODEVICE).
"""
import ctypes as C

import numpy as np

from . import _lib
from . import weights as W

COMP_ID = {"Luma": _lib.PMP_LUMA, "Chroma": _lib.PMP_CHROMA}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.uint8)


class Engine:
    def __init__(self, device=0, weight_dir=None, chunk=None, allow_synthetic_mtt=False):
        """allow_synthetic_mtt: let load() fall back to the synthetic MTT-net weights when a *_BD_* file is missing (tests,
        bench, smoke); off by default - a production run with a missing model file must fail (Inference_QBD.py:219-222)."""
        self.lib = _lib.load()
        self.allow_synthetic_mtt = bool(allow_synthetic_mtt)
        self.h = C.c_void_p()
        _lib.check(self.lib.pmp_create(int(device), C.byref(self.h)))
        self.device = device
        self.weight_dir = weight_dir
        self.provenance = {}
        if chunk:
            self.set_chunk(chunk)

    # ------------------------------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.pmp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        return _lib.check(rc, self.h)

    def set_stream(self, hip_stream_ptr):
        self._ck(self.lib.pmp_set_stream(self.h, C.c_void_p(hip_stream_ptr or 0)))

    def set_chunk(self, blocks):
        self._ck(self.lib.pmp_set_chunk(self.h, int(blocks)))

    def set_overlap(self, on):
        """Two chunks of a large call in flight on two streams (include/pmp.h: pmp_set_overlap)."""
        self._ck(self.lib.pmp_set_overlap(self.h, 1 if on else 0))

    def workspace_bytes(self):
        return int(self.lib.pmp_get_workspace_bytes(self.h))

    def set_precision(self, mode):
        """'f16x3' (default; 2-term fp16 split, 3 MFMA products), 'bf16x6' (3-term bf16 split, 6 products) - both
        fp32-equivalent - or 'fp32' (exact fp32 MFMA)."""
        self._ck(self.lib.pmp_set_precision(self.h, {"fp32": 0, "f32": 0, "bf16x6": 1, "f16x3": 2}[mode]))

    def set_fusion(self, on):
        """f16x3 launch fusion (include/pmp.h: pmp_debug_set_fusion; bit-identical results either way): True / 1 = all (default), False / 0 =
        launch per layer, 2 = only the 16x16 tails (chain16.hip), 3 = only the 32x32 ResidualBlocks (rbfuse32.hip)."""
        self._ck(self.lib.pmp_debug_set_fusion(self.h, int(on)))

    def get_precision(self):
        return {0: "fp32", 1: "bf16x6", 2: "f16x3"}[self.lib.pmp_get_precision(self.h)]

    def set_saturation_policy(self, policy):
        """f16x3 range guard (include/pmp.h): 'rerun' (default: a call whose activations left the fp16 range is run again on
        the exact fp32 MFMA datapath), 'error' (PMP_E_RANGE instead) or 'ignore' (no check, fully asynchronous device calls)."""
        self._ck(self.lib.pmp_set_saturation_policy(self.h, {"rerun": 0, "error": 1, "ignore": 2}[policy]))

    def saturated(self):
        """True if any inference call since clear_saturation() drove an f16x3 activation beyond +-65504."""
        return bool(self._ck(self.lib.pmp_get_saturation(self.h)))

    def saturation_reruns(self):
        return int(self.lib.pmp_get_saturation_reruns(self.h))

    def clear_saturation(self):
        self._ck(self.lib.pmp_clear_saturation(self.h))

    def set_activation_scales(self, on):
        """f16x3: use the calibrated activation scales of the MTT nets (default) or exponents of zero (include/pmp.h; the range-guard tests)."""
        self._ck(self.lib.pmp_debug_set_activation_scales(self.h, 1 if on else 0))

    def activation_report(self, comp, qp):
        """f16x3 activation scales of the MTT net of (comp, qp) and the calibration record behind them (include/pmp.h):
        {"exps": [e0..e4], "seg_amax": [..5..], "tensors": [(name, segment, max |value|), ...]}.  Loads the pair if necessary."""
        self.load(comp, qp)
        exps = (C.c_int * 5)()
        amax = (C.c_float * 5)()
        buf = C.create_string_buffer(16384)
        n = self._ck(self.lib.pmp_debug_activation_report(self.h, COMP_ID[comp], int(qp), exps, amax, buf, len(buf)))
        rows = [ln.split() for ln in buf.value.decode().splitlines()]
        assert len(rows) == n
        return {"exps": list(exps), "seg_amax": list(amax), "tensors": [(r[0], int(r[1]), float(r[2])) for r in rows]}

    def synchronize(self):
        self._ck(self.lib.pmp_synchronize(self.h))

    # ------------------------------------------------------------------------------------------ weights
    def load_pretrain_model(self, net, qp, tensors):
        """net in {Luma_Q, Luma_MSBD, Chroma_Q, Chroma_MSBD}; tensors {state_dict name: float32 ndarray}."""
        names = list(tensors.keys())
        arrs = [np.ascontiguousarray(tensors[k], dtype=np.float32).ravel() for k in names]
        blob = np.concatenate(arrs) if arrs else np.zeros(0, np.float32)
        descs = (_lib.TensorDesc * len(names))()
        off = 0
        keep = []
        for i, k in enumerate(names):
            shp = tuple(np.shape(tensors[k]))
            kb = k.encode()
            keep.append(kb)
            descs[i].name = kb
            descs[i].ndim = len(shp)
            for j in range(4):
                descs[i].shape[j] = int(shp[j]) if j < len(shp) else 0
            descs[i].offset = off
            off += arrs[i].size
        self._ck(self.lib.pmp_load_weights(self.h, _lib.NET_IDS[net], int(qp), _ptr(blob), descs, len(names)))

    def has_weights(self, net, qp):
        return bool(self.lib.pmp_has_weights(self.h, _lib.NET_IDS[net], int(qp)))

    def check_available(self, comp, qp):
        """Raise FileNotFoundError now if load(comp, qp) would: the driver loads weights lazily, next to the GPU's work, but a
        missing model file must stop the job before any output (Inference_QBD.py:219-222)."""
        for kind in ("Q", "MSBD"):
            net = "%s_%s" % (comp, kind)
            if not self.has_weights(net, qp):
                W.find_net_weights(net, qp, self.weight_dir, allow_synthetic=self.allow_synthetic_mtt)

    def load(self, comp, qp, q_weights=None, msbd_weights=None):
        """Load both nets of (comp, qp); missing dicts are resolved by weights.load_net_weights()."""
        for kind, given in (("Q", q_weights), ("MSBD", msbd_weights)):
            net = "%s_%s" % (comp, kind)
            if given is None:
                if self.has_weights(net, qp):
                    continue
                kind, path = W.find_net_weights(net, qp, self.weight_dir, allow_synthetic=self.allow_synthetic_mtt)
                if kind == "pmpw":      # the library's own reader: no copies through Python (pmp_load_weights_file)
                    self._ck(self.lib.pmp_load_weights_file(self.h, _lib.NET_IDS[net], int(qp), path.encode()))
                    self.provenance[(net, qp)] = path
                    continue
                given, src = W.load_net_weights(net, qp, self.weight_dir, allow_synthetic=self.allow_synthetic_mtt)
            else:
                src = "caller"
            self.load_pretrain_model(net, qp, given)
            self.provenance[(net, qp)] = src

    # ------------------------------------------------------------------------------------------ host API
    def inference_pre_QBD(self, comp, qp, block_y, block_u=None, block_v=None):
        """Metrics.py:387-419.  block_y u8[N,68,68] (+ block_u/v u8[N,34,34] for Chroma)
        -> qt f32[N,1,8,8], bt f32[N,3,16,16], dire f32[N,3,16,16]."""
        by, bu, bv = _u8(block_y), _u8(block_u), _u8(block_v)
        n = self._check_blocks(comp, by, bu, bv)
        self.load(comp, qp)
        qt = np.empty((n, 1, 8, 8), np.float32); bt = np.empty((n, 3, 16, 16), np.float32); dire = np.empty((n, 3, 16, 16), np.float32)
        self._ck(self.lib.pmp_infer(self.h, COMP_ID[comp], int(qp), _ptr(by), _ptr(bu), _ptr(bv), n, _ptr(qt), _ptr(bt), _ptr(dire)))
        return qt, bt, dire

    def post_process(self, qt, bt, dire, comp):
        """eli_structual_error + Map_to_Partition per block (Metrics.py:764-774 without the file).
        qt RAW logits f32[N,(1,)8,8]; returns hor, ver u8[N,16,16], qt_u8 u8[N,8,8], dire_i8 i8[N,3,16,16]."""
        qt = np.ascontiguousarray(qt, np.float32); bt = np.ascontiguousarray(bt, np.float32); dire = np.ascontiguousarray(dire, np.float32)
        n = qt.size // 64
        if qt.size != n * 64 or bt.size != n * 768 or dire.size != n * 768:
            raise ValueError("post_process: expected qt[N,8,8], bt[N,3,16,16], dire[N,3,16,16]")
        hor = np.empty((n, 16, 16), np.uint8); ver = np.empty((n, 16, 16), np.uint8)
        q8 = np.empty((n, 8, 8), np.uint8); d8 = np.empty((n, 3, 16, 16), np.int8)
        self._ck(self.lib.pmp_postprocess(self.h, COMP_ID[comp], _ptr(qt), _ptr(bt), _ptr(dire), n, _ptr(hor), _ptr(ver), _ptr(q8), _ptr(d8)))
        return hor, ver, q8, d8

    def seq_post_process(self, input_qt_batch, input_bt_batch, input_dire_batch, comp, sub_numfrm, width, height, save_path):
        """Metrics.py:764-774: post-process every block of a sequence and write the PartitionMat file."""
        n_expected = int(sub_numfrm) * (int(height) // 64) * (int(width) // 64)
        hor, ver, q8, d8 = self.post_process(input_qt_batch, input_bt_batch, input_dire_batch, comp)
        if hor.shape[0] != n_expected:
            raise ValueError("seq_post_process: %d blocks given, geometry needs %d" % (hor.shape[0], n_expected))
        if save_path is not None:
            write_partition_file(save_path, sub_numfrm, height, width, hor, ver, q8, d8)
        return hor, ver, q8, d8

    def infer_postprocess(self, comp, qp, block_y, block_u=None, block_v=None, want_logits=False):
        by, bu, bv = _u8(block_y), _u8(block_u), _u8(block_v)
        n = self._check_blocks(comp, by, bu, bv)
        self.load(comp, qp)
        hor = np.empty((n, 16, 16), np.uint8); ver = np.empty((n, 16, 16), np.uint8)
        q8 = np.empty((n, 8, 8), np.uint8); d8 = np.empty((n, 3, 16, 16), np.int8)
        qt = bt = dire = None
        if want_logits:
            qt = np.empty((n, 1, 8, 8), np.float32); bt = np.empty((n, 3, 16, 16), np.float32); dire = np.empty((n, 3, 16, 16), np.float32)
        self._ck(self.lib.pmp_infer_postprocess(self.h, COMP_ID[comp], int(qp), _ptr(by), _ptr(bu), _ptr(bv), n, _ptr(hor), _ptr(ver),
                                                _ptr(q8), _ptr(d8), _ptr(qt), _ptr(bt), _ptr(dire)))
        return (hor, ver, q8, d8, qt, bt, dire) if want_logits else (hor, ver, q8, d8)

    def output_block_yuv(self, y, u, v, bitdepth=8):
        """Inference_QBD.py:104-149 on already-loaded planes y[F,H,W], u,v[F,H/2,W/2] (u8, or u16 for 10-bit)."""
        dt = np.uint8 if bitdepth == 8 else np.uint16
        y = np.ascontiguousarray(y, dt); u = np.ascontiguousarray(u, dt); v = np.ascontiguousarray(v, dt)
        F, H, Wd = y.shape
        if u.shape != (F, H // 2, Wd // 2) or v.shape != u.shape:
            raise ValueError("output_block_yuv: chroma planes must be [F, H/2, W/2]")
        n = F * (H // 64) * (Wd // 64)
        by = np.empty((n, 68, 68), np.uint8); bu = np.empty((n, 34, 34), np.uint8); bv = np.empty((n, 34, 34), np.uint8)
        self._ck(self.lib.pmp_cut_blocks(self.h, _ptr(y), _ptr(u), _ptr(v), F, H, Wd, int(bitdepth), _ptr(by), _ptr(bu), _ptr(bv)))
        return by, bu, bv

    # ------------------------------------------------------------------------------------------ device API
    def infer_device(self, comp, qp, d_by, d_bu, d_bv, n, d_qt, d_bt, d_dire):
        self._ck(self.lib.pmp_infer_device(self.h, COMP_ID[comp], int(qp), d_by, d_bu, d_bv, int(n), d_qt, d_bt, d_dire))

    def postprocess_device(self, comp, d_qt, d_bt, d_dire, n, d_hor, d_ver, d_q8, d_d8):
        self._ck(self.lib.pmp_postprocess_device(self.h, COMP_ID[comp], d_qt, d_bt, d_dire, int(n), d_hor, d_ver, d_q8, d_d8))

    def infer_postprocess_device(self, comp, qp, d_by, d_bu, d_bv, n, d_hor, d_ver, d_q8, d_d8, d_qt=None, d_bt=None, d_dire=None):
        self._ck(self.lib.pmp_infer_postprocess_device(self.h, COMP_ID[comp], int(qp), d_by, d_bu, d_bv, int(n), d_hor, d_ver, d_q8,
                                                       d_d8, d_qt, d_bt, d_dire))

    def infer_postprocess_records_device(self, comp, qp, d_by, d_bu, d_bv, n, d_rec):
        """Blocks in, one packed 1344-byte record per block out (hor | ver | qt | dire): what the multi-GPU gather moves."""
        self._ck(self.lib.pmp_infer_postprocess_records_device(self.h, COMP_ID[comp], int(qp), d_by, d_bu, d_bv, int(n), d_rec))

    def postprocess_records_device(self, comp, d_qt, d_bt, d_dire, n, d_rec):
        self._ck(self.lib.pmp_postprocess_records_device(self.h, COMP_ID[comp], d_qt, d_bt, d_dire, int(n), d_rec))

    def cut_blocks_device(self, d_y, d_u, d_v, F, H, Wd, bitdepth, d_by, d_bu, d_bv):
        self._ck(self.lib.pmp_cut_blocks_device(self.h, d_y, d_u, d_v, int(F), int(H), int(Wd), int(bitdepth), d_by, d_bu, d_bv))

    # ------------------------------------------------------------------------------------------ timing
    def ktime_enable(self, mask):
        self._ck(self.lib.pmp_ktime_enable(self.h, int(mask)))

    def ktime(self):
        """{class name: (launches, total ms, algorithmic FLOPs)} accumulated since ktime_enable()."""
        out = {}
        for k in range(self.lib.pmp_ktime_classes()):
            n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
            self._ck(self.lib.pmp_ktime_get(self.h, k, C.byref(n), C.byref(ms), C.byref(fl)))
            out[self.lib.pmp_ktime_name(k).decode()] = (n.value, ms.value, fl.value)
        return out

    # ------------------------------------------------------------------------------------------ helpers
    @staticmethod
    def _check_blocks(comp, by, bu, bv):
        if comp not in COMP_ID:
            raise ValueError("comp must be 'Luma' or 'Chroma'")
        if by.ndim != 3 or by.shape[1:] != (68, 68):
            raise ValueError("block_y must be u8[N,68,68]")
        n = by.shape[0]
        if comp == "Chroma":
            if bu is None or bv is None or bu.shape != (n, 34, 34) or bv.shape != (n, 34, 34):
                raise ValueError("Chroma needs block_u and block_v u8[N,34,34]")
        return n


def write_partition_file(path, frames, height, width, hor, ver, qt_u8, dire_i8):
    """Map2Partition.py:385-412 (tiling + text emission) through the C writer; host-only, needs no GPU."""
    lib = _lib.load()
    hor = np.ascontiguousarray(hor, np.uint8); ver = np.ascontiguousarray(ver, np.uint8)
    q8 = np.ascontiguousarray(qt_u8, np.uint8); d8 = np.ascontiguousarray(dire_i8, np.int8)
    n = int(frames) * (int(height) // 64) * (int(width) // 64)
    if hor.size != n * 256 or ver.size != n * 256 or q8.size != n * 64 or d8.size != n * 768:
        raise ValueError("write_partition_file: array sizes do not match frames*(H//64)*(W//64) blocks")
    _lib.check(lib.pmp_write_partition_file(str(path).encode(), int(frames), int(height), int(width), _ptr(hor), _ptr(ver), _ptr(q8), _ptr(d8)))


def write_partition_binary(path, frames, height, width, hor, ver, qt_u8, dire_i8):
    """Binary side channel of the same data (include/pmp.h): frame matrices as the VTM keeps them."""
    lib = _lib.load()
    hor = np.ascontiguousarray(hor, np.uint8); ver = np.ascontiguousarray(ver, np.uint8)
    q8 = np.ascontiguousarray(qt_u8, np.uint8); d8 = np.ascontiguousarray(dire_i8, np.int8)
    n = int(frames) * (int(height) // 64) * (int(width) // 64)
    if hor.size != n * 256 or ver.size != n * 256 or q8.size != n * 64 or d8.size != n * 768:
        raise ValueError("write_partition_binary: array sizes do not match frames*(H//64)*(W//64) blocks")
    _lib.check(lib.pmp_write_partition_binary(str(path).encode(), int(frames), int(height), int(width), _ptr(hor), _ptr(ver), _ptr(q8), _ptr(d8)))


def tile_partition_maps(frames, height, width, hor, ver, qt_u8, dire_i8):
    """Per-block flags -> the frame matrices the patched VTM keeps after parsing (EncAppCfg.cpp:4270-4298): hor, ver
    u8[F][R][C], qt u8[F][R/2][C/2], dire i8[F][3][R][C], R = 16*(H>>6), C = 16*(W>>6) (pmp_tile_partition_maps)."""
    lib = _lib.load()
    R, Cc = 16 * (height // 64), 16 * (width // 64)
    hor, ver, qt_u8 = _u8(hor), _u8(ver), _u8(qt_u8)
    dire_i8 = np.ascontiguousarray(dire_i8, dtype=np.int8)
    oh = np.zeros((frames, R, Cc), np.uint8); ov = np.zeros_like(oh)
    oq = np.zeros((frames, R // 2, Cc // 2), np.uint8); od = np.zeros((frames, 3, R, Cc), np.int8)
    rc = lib.pmp_tile_partition_maps(frames, height, width, _ptr(hor), _ptr(ver), _ptr(qt_u8), _ptr(dire_i8), _ptr(oh), _ptr(ov),
                                     _ptr(oq), _ptr(od))
    if rc != 0:
        raise _lib.PmpError(rc, lib.pmp_last_error(None).decode())
    return oh, ov, oq, od


def read_partition_binary(path):
    """-> (frames, height, width, hor[F,R,C] u8, ver[F,R,C] u8, qt[F,R/2,C/2] u8, dire[F,3,R,C] i8) via numpy.memmap."""
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    if bytes(raw[:8]) != b"PMPB1\0\0\0":
        raise ValueError("%s: not a PMPB1 file" % path)
    frames, H, Wd, R, Cc = (int(v) for v in np.frombuffer(raw[8:28].tobytes(), dtype="<i4"))
    per = 5 * R * Cc + R * Cc // 4
    body = raw[40:40 + frames * per].reshape(frames, per)
    hor = body[:, :R * Cc].reshape(frames, R, Cc); ver = body[:, R * Cc:2 * R * Cc].reshape(frames, R, Cc)
    qt = body[:, 2 * R * Cc:2 * R * Cc + R * Cc // 4].reshape(frames, R // 2, Cc // 2)
    dire = body[:, 2 * R * Cc + R * Cc // 4:].reshape(frames, 3, R, Cc).view(np.int8)
    return frames, H, Wd, hor, ver, qt, dire


def format_partition_text(frames, height, width, hor, ver, qt_u8, dire_i8):
    lib = _lib.load()
    hor = np.ascontiguousarray(hor, np.uint8); ver = np.ascontiguousarray(ver, np.uint8)
    q8 = np.ascontiguousarray(qt_u8, np.uint8); d8 = np.ascontiguousarray(dire_i8, np.int8)
    need = _lib.check(lib.pmp_format_partition_text(int(frames), int(height), int(width), _ptr(hor), _ptr(ver), _ptr(q8), _ptr(d8), None, 0))
    buf = C.create_string_buffer(max(int(need), 1))
    got = _lib.check(lib.pmp_format_partition_text(int(frames), int(height), int(width), _ptr(hor), _ptr(ver), _ptr(q8), _ptr(d8), buf, need))
    assert got == need
    return buf.raw[:need]


def format_partition_rows_records(width, block_rows, rec):
    """Text of `block_rows` consecutive block rows of one frame from packed records u8[block_rows * (width // 64), 1344]
    (pmp_format_partition_rows_records): -> (ctypes char buffer holding the six sections back to back, int64[block_rows, 6] exact
    byte count of every (block row, section)).  Host-only; the C formatter releases the GIL."""
    lib = _lib.load()
    rec = np.ascontiguousarray(rec, np.uint8)
    bw = int(width) // 64
    if rec.size != int(block_rows) * bw * _lib.PMP_RECORD_BYTES:
        raise ValueError("format_partition_rows_records: %d bytes of records for %d block rows of %d blocks" % (rec.size, block_rows, bw))
    sizes = np.zeros((int(block_rows), 6), np.int64)
    need = _lib.check(lib.pmp_format_partition_rows_records(int(width), int(block_rows), _ptr(rec), None, 0, _ptr(sizes)))
    buf = C.create_string_buffer(max(int(need), 1))
    got = _lib.check(lib.pmp_format_partition_rows_records(int(width), int(block_rows), _ptr(rec), buf, need, None))
    assert got == need == int(sizes.sum())
    return buf, sizes


def tile_partition_rows_records(width, block_rows, rec):
    """The same rows as matrices: hor, ver u8[16 n, C], qt u8[8 n, C/2], dire i8[3, 16 n, C] (pmp_tile_partition_rows_records)."""
    lib = _lib.load()
    rec = np.ascontiguousarray(rec, np.uint8)
    R, Cc = 16 * int(block_rows), 16 * (int(width) // 64)
    if rec.size != int(block_rows) * (Cc // 16) * _lib.PMP_RECORD_BYTES:
        raise ValueError("tile_partition_rows_records: record count does not match the geometry")
    oh = np.zeros((R, Cc), np.uint8); ov = np.zeros_like(oh); oq = np.zeros((R // 2, Cc // 2), np.uint8); od = np.zeros((3, R, Cc), np.int8)
    if rec.size:
        _lib.check(lib.pmp_tile_partition_rows_records(int(width), int(block_rows), _ptr(rec), _ptr(oh), _ptr(ov), _ptr(oq), _ptr(od)))
    return oh, ov, oq, od


def read_partition_file(path, frames, height, width):
    """Parser with the geometry rules of EncAppCfg::parsePartitionMatrix (EncAppCfg.cpp:4247-4250, :4299-4399):
    returns hor, ver [F,R,C], qt [F,R/2,C/2], dire [F,3,R,C] as int arrays (frame matrices, not per block)."""
    vals = np.array(open(path).read().split(), dtype=np.int64)
    R, Cc = (height >> 6) * 16, (width >> 6) * 16
    per = 5 * R * Cc + R * Cc // 4
    if vals.size != frames * per:
        raise ValueError("%s: %d values, expected %d" % (path, vals.size, frames * per))
    v = vals.reshape(frames, per)
    hor = v[:, :R * Cc].reshape(frames, R, Cc); ver = v[:, R * Cc:2 * R * Cc].reshape(frames, R, Cc)
    qt = v[:, 2 * R * Cc:2 * R * Cc + R * Cc // 4].reshape(frames, R // 2, Cc // 2)
    dire = v[:, 2 * R * Cc + R * Cc // 4:].reshape(frames, 3, R, Cc)
    return hor, ver, qt, dire
