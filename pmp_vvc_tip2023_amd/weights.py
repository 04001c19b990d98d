"""Weight containers for the four Down-Up-CNN nets.

The reference stores weights as legacy torch pickles with CUDA-tagged storages and DataParallel `module.`
prefixes (trained_models/*.pkl; loader Inference_QBD.py:28-46).  The product's own container is `.pmpw`:

    b"PMPW1\\n" | u32 little-endian JSON length | JSON manifest | raw little-endian float32 payload

manifest = {"net": "Luma_Q", "qp": 22, "source": ..., "tensors": [{"name", "shape", "offset"(floats)}]
            [, "act_exp": [5 ints], "act_fp": [fingerprint of these tensors, of the QT partner's: 16 hex digits each]]}.
Tensor names/shapes are the reference's state_dict names with `module.` stripped (OIHW convs, 1-D biases).
`.pmpw` needs only numpy; `.pkl` import needs torch (PyTorch is used for weight loading only).
"""
import json
import os
import struct

import numpy as np

MAGIC = b"PMPW1\n"
NETS = ("Luma_Q", "Luma_MSBD", "Chroma_Q", "Chroma_MSBD")
QPS = (22, 27, 32, 37)


ACT_EXP_MAX, ACT_EXP_ATT_MAX = 30, 6      # csrc/pmp_hostonly.h: bounds of a manifest's exponents (trunk segments 0, 2, 4 / attention segments 1, 3)
_M64 = (1 << 64) - 1


def fingerprint(tensors):
    """csrc/pmpw_file.cpp: fingerprint_tensors - the same 64-bit number from {name: ndarray}: tensors sorted by name; FNV-1a over the name,
    the shape folded in, a position-weighted sum of the float32 bit patterns."""
    P = 0x100000001b3
    fp = 0xcbf29ce484222325
    for name in sorted(tensors, key=lambda k: k.encode()):
        a = np.ascontiguousarray(tensors[name], dtype="<f4")
        h = 0xcbf29ce484222325
        for ch in name.encode():
            h = ((h ^ ch) * P) & _M64
        for d in a.shape:
            h = ((h ^ int(d)) * P) & _M64
        bits = a.reshape(-1).view("<u4").astype(np.uint64)
        with np.errstate(over="ignore"):
            s = int(((bits + np.uint64(0x9E3779B97F4A7C15)) * (np.arange(bits.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))).sum(dtype=np.uint64))
        h = ((h ^ s) * P) & _M64
        fp = ((fp ^ h) * P) & _M64
    return fp


def save_pmpw(path, net, qp, tensors, source="", act_exp=None, qt_partner=None):
    """tensors: ordered {name: float32 ndarray}.  act_exp (MTT nets, optional): the five f16x3 activation-scale exponents a calibration on
    the target GPU chose (Engine.activation_report(...)["exps"]); a file that carries them is loaded without a calibration pass.
    qt_partner (with act_exp): the QT net's tensors the calibration ran with - their fingerprint and this net's go into "act_fp", and the
    library ignores the exponents when either does not match what is loaded (a stale file)."""
    entries, off = [], 0
    for name, a in tensors.items():
        a = np.ascontiguousarray(a, dtype="<f4")
        entries.append({"name": name, "shape": list(a.shape), "offset": off})
        off += a.size
    man = {"net": net, "qp": int(qp), "source": source, "tensors": entries}
    if act_exp is not None:
        if len(act_exp) != 5 or any(int(e) < 0 or int(e) > (ACT_EXP_ATT_MAX if i in (1, 3) else ACT_EXP_MAX) for i, e in enumerate(act_exp)):
            raise ValueError("act_exp: five integers, 0..%d for segments 0, 2, 4 and 0..%d for the attention segments 1, 3" % (ACT_EXP_MAX, ACT_EXP_ATT_MAX))
        man["act_exp"] = [int(e) for e in act_exp]
        if qt_partner is not None:
            man["act_fp"] = ["%016x" % fingerprint(tensors), "%016x" % fingerprint(qt_partner)]
    man = json.dumps(man).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(man)))
        f.write(man)
        for a in tensors.values():
            f.write(np.ascontiguousarray(a, dtype="<f4").tobytes())


def load_pmpw(path):
    """-> (manifest dict, {name: float32 ndarray})."""
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError("%s: not a PMPW1 file" % path)
        (n,) = struct.unpack("<I", f.read(4))
        man = json.loads(f.read(n).decode())
        payload = np.frombuffer(f.read(), dtype="<f4")
    out = {}
    for e in man["tensors"]:
        cnt = int(np.prod(e["shape"])) if e["shape"] else 1
        out[e["name"]] = payload[e["offset"]:e["offset"] + cnt].reshape(e["shape"]).astype(np.float32)
    return man, out


def load_pkl(path):
    """Reference pickle -> {name: float32 ndarray}; mirrors remove_prefix/load_pretrain_model
    (Inference_QBD.py:28-46) but with map_location='cpu' (the shipped loader fails on CPU-only hosts)."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=True)   # plain state_dicts: no pickle code is executed
    if "state_dict" in sd:
        sd = sd["state_dict"]
    return {(k.split("module.", 1)[-1] if k.startswith("module.") else k):
            v.detach().float().contiguous().numpy() for k, v in sd.items()}


def default_weight_dir():
    return os.environ.get("PMP_WEIGHT_DIR",
                          os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights"))


def ref_net_name(net):
    """Our net id -> the reference's file stem: Luma_Q -> Luma_Q, Luma_MSBD -> Luma_BD (trained_models/README.md)."""
    comp, kind = net.split("_", 1)
    return comp + ("_Q" if kind == "Q" else "_BD")


def find_net_weights(net, qp, weight_dir=None, allow_synthetic=False):
    """Where the weights of (net, qp) will come from, without reading them: ("pmpw" | "pkl", path) or ("synthetic", None).
    Resolution order: <dir>/<Comp>_{Q,BD}_<qp>.pmpw, then .pkl (reference naming, Inference_QBD.py:219-220).  A missing file is an
    error, as in the reference (Inference_QBD.py:219-222 dies on a missing model file) - unless allow_synthetic=True, which for the
    MTT nets only (their files are absent from the reference mount) falls back to the documented synthetic generator
    (seed = qp): tests, bench.py and smoke() ask for it explicitly, the CLI driver only with --allowSyntheticMTT."""
    d = weight_dir or default_weight_dir()
    stem = "%s_%d" % (ref_net_name(net), qp)
    for kind in ("pmpw", "pkl"):
        p = os.path.join(d, stem + "." + kind)
        if os.path.isfile(p):
            return kind, p
    if net.endswith("_MSBD") and allow_synthetic:
        return "synthetic", None
    hint = " (MTT-net files are not part of the reference checkout; --allowSyntheticMTT / allow_synthetic=True runs on " \
           "documented synthetic weights instead)" if net.endswith("_MSBD") else ""
    raise FileNotFoundError("no weights for %s qp%d: neither %s.pmpw nor %s.pkl under %s%s" % (net, qp, stem, stem, d, hint))


def load_net_weights(net, qp, weight_dir=None, allow_synthetic=False):
    """Weights of (net, qp) as find_net_weights resolves them.  Returns (weights, provenance-string)."""
    kind, p = find_net_weights(net, qp, weight_dir, allow_synthetic)
    if kind == "pmpw":
        return load_pmpw(p)[1], p
    if kind == "pkl":
        return load_pkl(p), p
    from . import synth
    return synth.synth_msbd_weights(net.split("_")[0], qp), "synthetic(seed=%d)" % qp
