"""Generates tests/golden/regionprops_skimage_general.npz with the scikit-image of the build container (0.18.3 under
/opt/conda/bin/python3.9):

    cd /root/repo && /opt/conda/bin/python3.9 tests/golden/make_golden_skimage_general.py

General label images for calculate_regionprops (detection.py:75-159) -- what regionprops(mask, intensity_image=img)[0]
returns when the region does NOT span the window, has background holes, or the mask holds several label values
(props[0] is the region of the smallest one).  Data only: masks, intensity images and the 32 expected columns."""
import numpy as np
from scipy import ndimage as ndi

from make_golden_skimage import KEYS, props32


def main():
    rng = np.random.RandomState(20261111)
    masks, imgs, shapes, exp = [], [], [], []
    for i in range(30):
        h, w = rng.randint(20, 140, 2)
        f = ndi.gaussian_filter(rng.standard_normal((h, w)), rng.uniform(1.5, 5.0))
        m = f > np.quantile(f, rng.uniform(0.5, 0.85))
        m[:rng.randint(0, 6)] = False                       # empty margins: the bounding box is offset inside the window
        m[:, :rng.randint(0, 9)] = False
        m[h - rng.randint(0, 5):] = False
        if m.sum() < 20:
            m[h // 3:h // 2 + 2, w // 3:w // 2 + 2] = True
        mask = np.where(m, 255, 0).astype(np.uint8)
        if i % 3 == 1:                                      # a second, smaller label value: it is the one regionprops()[0] describes
            lab, n = ndi.label(m, structure=np.ones((3, 3)))
            k = 1 + int(np.argmax(ndi.sum(m, lab, range(1, n + 1))))
            mask[lab == k] = 90
        if i % 3 == 2:                                      # punch holes
            yy, xx = np.nonzero(m)
            for j in rng.choice(len(yy), min(6, len(yy)), replace=False):
                mask[max(yy[j] - 1, 0):yy[j] + 1, max(xx[j] - 1, 0):xx[j] + 2] = 0
        if not (mask > 0).any():
            mask[h // 2, w // 2] = 255
        img = rng.randint(0, 256, mask.shape).astype(np.uint8)
        masks.append(mask.ravel()); imgs.append(img.ravel()); shapes.append(mask.shape)
        exp.append(props32(mask, img))
    np.savez_compressed("tests/golden/regionprops_skimage_general.npz", keys=np.array(KEYS), shapes=np.array(shapes, np.int32),
                        masks=np.concatenate(masks), imgs=np.concatenate(imgs), expected=np.array(exp, np.float64))
    print("cases", len(exp))


if __name__ == "__main__":
    main()
