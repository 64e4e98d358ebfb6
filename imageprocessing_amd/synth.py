"""Synthetic PISCO-like frame stacks.  The object tables are drawn on the
host; pixels are rendered either here in NumPy (small stacks, CPU tests, CPU baseline) or on the
device by ``Segmenter.synth_frames`` -- both follow csrc/synth_core.h and give identical bytes.
The reference has no data generator; this exists so that every run uses the same inputs."""
import numpy as np

M64 = (1 << 64) - 1


def _hash(seed, frame, pix):
    """synth_hash of csrc/synth_core.h on uint64 arrays."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(frame) * np.uint64(0xBF58476D1CE4E5B9)
             + pix.astype(np.uint64) * np.uint64(0x94D049BB133111EB) + np.uint64(0x2545F4914F6CDD1D))
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def make_objects(n_frames, H, W, seed, per_frame=50, dmin=8.0, dmax=300.0, ring_frac=0.2, frame0=0):
    """Object table int32[n_obj, 8] = cx, cy, a, b, cosQ12, sinQ12, contrast, kind and offsets [n+1].
    Diameters are log-uniform in [dmin, dmax] scaled by W/2560; every object lives in one frame only."""
    scale = W / 2560.0
    objs, off = [], [0]
    for f in range(frame0, frame0 + n_frames):
        rng = np.random.default_rng([seed, f])
        k = per_frame
        d = np.exp(rng.uniform(np.log(dmin), np.log(dmax), k)) * scale
        asp = rng.uniform(1.0, 3.0, k)
        a = np.clip(np.round(d / 2 * np.sqrt(asp)), 1, 400).astype(np.int64)
        b = np.clip(np.round(d / 2 / np.sqrt(asp)), 1, 400).astype(np.int64)
        th = rng.uniform(0, np.pi, k)
        rr = 1150.0 * scale * np.sqrt(rng.uniform(0, 1, k))
        ph = rng.uniform(0, 2 * np.pi, k)
        cx = np.round(W / 2 + rr * np.cos(ph)).astype(np.int64)
        cy = np.round(H / 2 + rr * np.sin(ph)).astype(np.int64)
        contrast = rng.integers(30, 121, k)
        u = rng.uniform(0, 1, k)
        kind = np.where((u < ring_frac) & (b >= 6), np.where(u < ring_frac / 2, 2, 1), 0)
        o = np.stack([cx, cy, a, b, np.round(np.cos(th) * 4096).astype(np.int64), np.round(np.sin(th) * 4096).astype(np.int64),
                      contrast, kind], axis=1).astype(np.int32)
        objs.append(o)
        off.append(off[-1] + k)
    return (np.concatenate(objs) if objs else np.zeros((0, 8), np.int32)), np.asarray(off, np.int32)


def render(n_frames, H, W, seed, objs, off, frame0=0):
    """NumPy mirror of k_synth: uint8[n, H, W]."""
    out = np.empty((n_frames, H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.int64)
    r2 = (xx - W // 2) ** 2 + (yy - H // 2) ** 2
    rin, rout = (1230 * W) // 2560, (1240 * W) // 2560
    rin2, rout2 = rin * rin, rout * rout
    base0 = np.where(r2 <= rin2, 170 - (15 * r2) // rin2, np.where(r2 >= rout2, 30, 155 - (125 * (r2 - rin2)) // (rout2 - rin2)))
    pix = (yy * W + xx).astype(np.uint64)
    for f in range(n_frames):
        fr = frame0 + f
        g = 990 + int(_hash(seed, fr, np.array([0xFFFFFFFF], np.uint64))[0] % np.uint64(21))
        z = _hash(seed, fr, pix)
        t = ((z & np.uint64(255)) + ((z >> np.uint64(8)) & np.uint64(255)) + ((z >> np.uint64(16)) & np.uint64(255))
             + ((z >> np.uint64(24)) & np.uint64(255))).astype(np.int64) * 3
        v = base0 * g // 1000 + t // 255 - 6
        for o in objs[off[f]:off[f + 1]].astype(np.int64):
            cx, cy, a, b, cq, sq, con, kind = o
            ext = max(a, b) + 1
            x0, x1, y0, y1 = max(cx - ext, 0), min(cx + ext, W - 1), max(cy - ext, 0), min(cy + ext, H - 1)
            if x0 > x1 or y0 > y1:
                continue
            dx = xx[y0:y1 + 1, x0:x1 + 1] - cx
            dy = yy[y0:y1 + 1, x0:x1 + 1] - cy
            u = dx * cq + dy * sq
            w = -dx * sq + dy * cq
            e = u * u * b * b + w * w * a * a
            lim = a * a * b * b * (1 << 24)
            hit = e <= lim
            if kind != 0:
                body = e > lim // 4
                if kind == 2:
                    body |= e <= lim // 25
                hit &= body
            v[y0:y1 + 1, x0:x1 + 1] -= con * hit
        out[f] = np.clip(v, 0, 255).astype(np.uint8)
    return out


CONFIGS = {
    # BASELINE.json configs (object statistics only; frame counts are chosen by the caller)
    "pisco": dict(per_frame=50, dmin=8.0, dmax=300.0, ring_frac=0.2),
    "uvp5_dense": dict(per_frame=2000, dmin=2.0, dmax=12.0, ring_frac=0.0),     # configs[3]: every object below the save threshold
    # the same density with larger particles: ~550 saved objects per frame (area > 400), overlapping bounding boxes -- the
    # per-object crop / property / hull kernels and the cumulative mask under density
    "uvp5_dense_saved": dict(per_frame=2000, dmin=8.0, dmax=40.0, ring_frac=0.1),
}
