"""Generates tests/golden/regionprops_skimage.npz with the scikit-image available in the build
container (0.18.3 under /opt/conda/bin/python3.9):

    cd /root/repo && /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py

For every random (mask, intensity) window it stores the 32 columns that the reference's
calculate_regionprops (detection.py:75-159) reads from skimage.measure.regionprops(mask,
intensity_image=img)[0].  The >=0.19 attribute names intensity_min/max/mean used by the reference
are the 0.18 names min_/max_/mean_intensity.  Data only: no reference source is stored.
"""
import numpy as np
from scipy import ndimage as ndi
from skimage import measure
import skimage

KEYS = ["width", "height", "bx", "by", "circ.", "area_exc", "area_rprops", "%area", "major", "minor",
        "centroid_y", "centroid_x", "convex_area", "min_intensity", "max_intensity", "mean_intensity",
        "intden", "perim.", "elongation", "range", "perimareaexc", "perimmajor", "circex", "angle",
        "bounding_box_area", "eccentricity", "equivalent_diameter", "euler_number", "extent",
        "local_centroid_col", "local_centroid_row", "solidity"]


def props32(mask, img):
    p = measure.regionprops(mask, intensity_image=img)[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        d = {
            "width": p.bbox[3] - p.bbox[1], "height": p.bbox[2] - p.bbox[0], "bx": p.bbox[1], "by": p.bbox[0],
            "circ.": (4 * np.pi * p.filled_area) / p.perimeter ** 2,
            "area_exc": p.area, "area_rprops": p.filled_area, "%area": 1 - (p.area / p.filled_area),
            "major": p.major_axis_length, "minor": p.minor_axis_length,
            "centroid_y": p.centroid[0], "centroid_x": p.centroid[1], "convex_area": p.convex_area,
            "min_intensity": p.min_intensity, "max_intensity": p.max_intensity, "mean_intensity": p.mean_intensity,
            "intden": p.filled_area * p.mean_intensity, "perim.": p.perimeter,
            "elongation": np.divide(p.major_axis_length, p.minor_axis_length),
            "range": p.max_intensity - p.min_intensity,
            "perimareaexc": p.perimeter / p.area, "perimmajor": p.perimeter / p.major_axis_length,
            "circex": np.divide(4 * np.pi * p.area, p.perimeter ** 2),
            "angle": p.orientation / np.pi * 180 + 90,
            "bounding_box_area": p.bbox_area, "eccentricity": p.eccentricity,
            "equivalent_diameter": p.equivalent_diameter, "euler_number": p.euler_number, "extent": p.extent,
            "local_centroid_col": p.local_centroid[1], "local_centroid_row": p.local_centroid[0],
            "solidity": p.solidity,
        }
    return [float(d[k]) for k in KEYS]


def blob(rng, h, w, kind):
    """window whose 255-pixels touch all four sides (like a c_mask cut at the contour's bbox)"""
    while True:
        f = ndi.gaussian_filter(rng.standard_normal((h + 16, w + 16)), rng.uniform(1.5, 6.0))
        m = f > np.quantile(f, rng.uniform(0.45, 0.8))
        lab, n = ndi.label(m, structure=np.ones((3, 3)))
        if n == 0:
            continue
        sizes = ndi.sum(m, lab, range(1, n + 1))
        k = int(np.argmax(sizes)) + 1
        sl = ndi.find_objects(lab)[k - 1]
        comp = lab[sl] == k
        if kind == 0:                      # hole-free, single piece (the common c_mask)
            out = ndi.binary_fill_holes(comp)
        elif kind == 1:                    # holes kept (regionprops must still agree)
            out = comp
        else:                              # window also contains pieces of other objects (cumulative mask, F8)
            out = m[sl]
            if not (out[0].any() and out[-1].any() and out[:, 0].any() and out[:, -1].any()):
                continue
        if out.sum() >= 30 and min(out.shape) >= 4:
            return out


def main():
    rng = np.random.RandomState(20251212)
    masks, imgs, shapes, exp = [], [], [], []
    for i in range(160):
        h, w = rng.randint(12, 150, 2)
        m = blob(rng, h, w, i % 3)
        img = rng.randint(0, 256, m.shape).astype(np.uint8)
        if i % 5 == 0:
            img[:] = 200
        mask = (m * 255).astype(np.uint8)
        crop_masked = np.where(mask == 255, img, 255).astype(np.uint8)
        exp.append(props32(mask, crop_masked))
        masks.append(np.packbits(m.ravel()))
        imgs.append(crop_masked.ravel())
        shapes.append(m.shape)
    # degenerate shapes
    for m in [np.ones((1, 7), bool), np.ones((5, 1), bool), np.ones((2, 2), bool), np.eye(6, dtype=bool),
              np.ones((9, 9), bool)]:
        img = rng.randint(0, 256, m.shape).astype(np.uint8)
        mask = (m * 255).astype(np.uint8)
        crop_masked = np.where(mask == 255, img, 255).astype(np.uint8)
        exp.append(props32(mask, crop_masked))
        masks.append(np.packbits(m.ravel()))
        imgs.append(crop_masked.ravel())
        shapes.append(m.shape)
    np.savez_compressed("tests/golden/regionprops_skimage.npz", masks=np.concatenate(masks), imgs=np.concatenate(imgs),
                        shapes=np.asarray(shapes, np.int32), expected=np.asarray(exp, np.float64),
                        keys=np.asarray(KEYS), skimage_version=np.asarray(skimage.__version__))
    print("wrote", len(shapes), "cases with skimage", skimage.__version__)


if __name__ == "__main__":
    main()
