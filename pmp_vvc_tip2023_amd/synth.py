"""Deterministic synthetic inputs and synthetic MTT-net weights.

Why this exists: the reference ships only the QT-net weights
(`trained_models/*_Q_*.pkl`); all eight `*_BD_*.pkl` MTT-net files and every test YUV are missing from the
mount.  Throughput runs and MTT parity therefore use
  * recipe R blocks: smooth random content (white noise drives Luma_Q_22 to a constant 1.0 output), and
  * MTT-net weights drawn from a documented PRNG (SplitMix64 -> uniform), shaped like the real tensors
    (Model_QBD.py:101-125 / :199-223) and scaled so the heads land in the range of real depth/direction maps.
Everything here is numpy-only so it runs identically in the fixture generator, the tests and bench.py.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------- PRNG
_M64 = (1 << 64) - 1


def splitmix64(seed, n):
    """n 64-bit outputs of SplitMix64 started at `seed` (vectorised; exact integer arithmetic)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & _M64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(seed, n):
    """float32 uniform in [-1, 1) with 24 random bits per value."""
    z = splitmix64(seed, n)
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 23) - 1.0).astype(np.float32)


def _name_seed(seed, name):
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & _M64
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & _M64


# ----------------------------------------------------------------------------------- MTT-net tensor list
def _resblock(prefix, cin, cout, k):
    t = [(prefix + ".left.0.weight", (cout, cin, k, k)), (prefix + ".left.2.weight", (cout, cout, k, k))]
    if cin != cout:
        t.append((prefix + ".shortcut.0.weight", (cout, cin, 1, 1)))
    return t


def msbd_tensor_shapes(comp):
    """Names/shapes of {Luma,Chroma}_MSBD_Net.state_dict() (Model_QBD.py:101-125, :199-223), in module order."""
    luma = comp == "Luma"
    cin = 2 if luma else 4
    k1, k2, k3 = ((9, 9), (5, 9), (9, 5)) if luma else ((5, 5), (3, 5), (5, 3))
    t = [("conv_b1_1.weight", (16, cin) + k1), ("conv_b1_1.bias", (16,)),
         ("conv_b1_2.weight", (8, cin) + k2), ("conv_b1_2.bias", (8,)),
         ("conv_b1_3.weight", (8, cin) + k3), ("conv_b1_3.bias", (8,))]
    t += _resblock("trunk_M1.0", 32, 64, 5)
    for i in range(1, 6):
        t += _resblock("trunk_M1.%d" % i, 64, 64, 3)
    for i in range(4):
        t += _resblock("trunk_M2.%d" % i, 64, 64, 3)
    for b in ("B1", "B2", "B3"):
        t += _resblock("trunk_%s.0" % b, 64, 32, 3) + _resblock("trunk_%s.1" % b, 32, 16, 3) + \
             _resblock("trunk_%s.2" % b, 16, 8, 3)
    for b in ("B1", "B2", "B3"):
        t += [("conv_%s.weight" % b, (2, 8, 3, 3)), ("conv_%s.bias" % b, (2,))]
    for a in ("Att1", "Att2"):
        t += _resblock("trunk_%s.0" % a, 3, 32, 3) + _resblock("trunk_%s.1" % a, 32, 64, 3)
    return t


def synth_msbd_weights(comp, seed):
    """Deterministic MTT-net weights: U(-b, b), b = g/sqrt(fan_in) per tensor (torch's default bound is g=1).

    Gains g (chosen so the heads spread over the value ranges Map2Partition.py works on; measured with the
    reference modules): stems see raw 0..255 pixels (Inference_QBD.py:195, no normalisation) -> 1/64;
    main trunks M1/M2: 1.0 / 0.5 / 1.0 for left.0 / left.2 / shortcut; branch and attention trunks (they
    shrink channels 64->8): 2.4 / 1.2 / 1.7; heads conv_B*: 6.0 with biases (0.5, 0.0) + 0.05*u.
    """
    out = {}
    for name, shape in msbd_tensor_shapes(comp):
        n = int(np.prod(shape))
        u = uniform_pm1(_name_seed(seed, comp + "/" + name), n).reshape(shape)
        if name.endswith(".bias"):
            if name.startswith("conv_b1_"):
                w = 0.1 * u
            else:  # heads: ch0 depth, ch1 direction
                w = (np.array([0.5, 0.0], np.float32) + 0.05 * u).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            if name.startswith("conv_b1_"):
                g = 1.0 / 64.0
            elif name.startswith("conv_B"):
                g = 6.0
            elif name.startswith("trunk_M"):
                g = 0.5 if ".left.2." in name else 1.0
            else:
                g = 2.4 if ".left.0." in name else (1.2 if ".left.2." in name else 1.7)
            w = (u * np.float32(g / np.sqrt(fan_in))).astype(np.float32)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


# --------------------------------------------------------------------------------------------- recipe R
def _bilinear_up4(grid):
    """(h+1, w+1) grid -> (4h, 4w) bilinear samples (pixel centres), float64."""
    h, w = grid.shape[0] - 1, grid.shape[1] - 1
    ys = (np.arange(4 * h) + 0.5) / 4.0
    xs = (np.arange(4 * w) + 0.5) / 4.0
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    g = grid.astype(np.float64)
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx


def recipe_r_plane(rng, h, w, sigma=6.0):
    """SURVEY.md 8(d) recipe R: bilinear x4 upsample of U{0..255} on a coarse grid + N(0,sigma), u8."""
    gh, gw = (h + 3) // 4, (w + 3) // 4
    grid = rng.integers(0, 256, size=(gh + 1, gw + 1)).astype(np.float64)
    img = _bilinear_up4(grid)[:h, :w] + rng.normal(0.0, sigma, size=(h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def recipe_r_blocks(n, seed):
    """n independent luma blocks u8[n,68,68] plus chroma u,v u8[n,34,34] (per-block content)."""
    rng = np.random.default_rng(seed)
    y = np.stack([recipe_r_plane(rng, 68, 68) for _ in range(n)])
    u = np.stack([recipe_r_plane(rng, 34, 34, sigma=3.0) for _ in range(n)])
    v = np.stack([recipe_r_plane(rng, 34, 34, sigma=3.0) for _ in range(n)])
    return y, u, v


def recipe_r_frames(nfrm, h, w, seed, bitdepth=8):
    """Planar 4:2:0 frames: y[F,H,W], u,v[F,H/2,W/2]; u8 or (10-bit) u16 holding 0..1023."""
    rng = np.random.default_rng(seed)
    y = np.stack([recipe_r_plane(rng, h, w) for _ in range(nfrm)])
    u = np.stack([recipe_r_plane(rng, h // 2, w // 2, sigma=3.0) for _ in range(nfrm)])
    v = np.stack([recipe_r_plane(rng, h // 2, w // 2, sigma=3.0) for _ in range(nfrm)])
    if bitdepth == 10:
        def up(p):
            return (p.astype(np.uint16) * 4 + rng.integers(0, 4, size=p.shape).astype(np.uint16))
        return up(y), up(u), up(v)
    return y, u, v


# ------------------------------------------------------------------- synthetic partition maps (post-proc)
def _split(x, y, h, w, mode):
    """Geometry of Map2Partition.py:124-138 (1 BT-H, 2 BT-V, 3 TT-H, 4 TT-V) with per-part depth increments."""
    if mode == 1:
        return [(x, y, h // 2, w, 1), (x + h // 2, y, h // 2, w, 1)]
    if mode == 2:
        return [(x, y, h, w // 2, 1), (x, y + w // 2, h, w // 2, 1)]
    if mode == 3:
        return [(x, y, h // 4, w, 2), (x + h // 4, y, h // 2, w, 1), (x + (h * 3) // 4, y, h // 4, w, 2)]
    return [(x, y, h, w // 4, 2), (x, y + w // 4, h, w // 2, 1), (x, y + (w * 3) // 4, h, w // 4, 2)]


def random_partition_maps(rng, cf=1, p_qt=0.55, p_mtt=0.6):
    """One random *valid* QT+MTT partition rendered the way the nets are trained to predict it:
    qt f32[8,8] (depth 0..3, constant on 2x2 so it survives eli_structual_error), bt f32[3,16,16] cumulative
    MTT depth after layers 1..3, dire f32[3,16,16] in {-1,0,1} (1 horizontal, -1 vertical)."""
    qt = np.zeros((8, 8), np.float32)
    bt = np.zeros((3, 16, 16), np.float32)
    dire = np.zeros((3, 16, 16), np.float32)

    def mtt(x, y, h, w):
        cus, cur = [(x, y, h, w)], np.zeros((16, 16), np.float32)
        for layer in range(3):
            nxt = []
            for (cx, cy, ch, cw) in cus:
                legal = []
                if ch >= 2 * cf and ch % (2 * cf) == 0: legal.append(1)
                if cw >= 2 * cf and cw % (2 * cf) == 0: legal.append(2)
                if ch >= 4 * cf and ch % (4 * cf) == 0: legal.append(3)
                if cw >= 4 * cf and cw % (4 * cf) == 0: legal.append(4)
                if not legal or rng.random() > p_mtt:
                    nxt.append((cx, cy, ch, cw))
                    continue
                mode = legal[rng.integers(len(legal))]
                dire[layer, cx:cx + ch, cy:cy + cw] = 1.0 if mode in (1, 3) else -1.0
                for (sx, sy, sh, sw, inc) in _split(cx, cy, ch, cw, mode):
                    cur[sx:sx + sh, sy:sy + sw] += inc
                    nxt.append((sx, sy, sh, sw))
            cus = nxt
            bt[layer, x:x + h, y:y + w] = cur[x:x + h, y:y + w]

    def quad(depth, qx, qy):
        sms = 8 >> depth
        if depth < 2 and rng.random() < p_qt or depth == 2 and rng.random() < p_qt * 0.6:
            for io in range(2):
                for jo in range(2):
                    quad(depth + 1, qx + io * sms // 2, qy + jo * sms // 2)
        else:
            qt[qx:qx + sms, qy:qy + sms] = depth
            mtt(2 * qx, 2 * qy, 2 * sms, 2 * sms)

    quad(0, 0, 0)
    return qt, bt, dire


def random_partition_batch(n, seed, cf=1, sigma=0.15):
    """n map triples: valid random partitions + N(0, sigma) on bt/dire (post-proc distribution).
    qt stays integral (it stands for the eli_structual_error output)."""
    rng = np.random.default_rng(seed)
    qt = np.zeros((n, 8, 8), np.float32); bt = np.zeros((n, 3, 16, 16), np.float32); dr = np.zeros_like(bt)
    for i in range(n):
        qt[i], bt[i], dr[i] = random_partition_maps(rng, cf)
    if sigma > 0:
        bt += rng.normal(0, sigma, bt.shape).astype(np.float32)
        dr += rng.normal(0, sigma, dr.shape).astype(np.float32)
    return qt, bt, dr
