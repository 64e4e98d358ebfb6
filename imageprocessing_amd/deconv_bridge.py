"""GPU-resident hand-off between the hot path and the deconvolution network.

The reference moves every frame through two queues and the host around LUCYD: ``x = cleaned / 255`` in NumPy float64,
``torch.from_numpy(x).to(device)``, ``model(batch.float())``, ``deconv * 255`` and ``astype(np.uint8)`` on the host
(deconvolution.py:70-100), then detection on ``(bg_corr_img, deconvolved)`` (segmenter.py:243-254; the Crops/ plane is
cut from the background-corrected image, Deconv_crops/ and the properties from the deconvolved one, detection.py:243-258).
Here the frames never leave the device: ``pseg_bgcorrect`` -> these tensor conversions -> any torch module ->
``pseg_detect(cleaned=deconvolved, raw=bg_corrected)``.  The network itself stays stock PyTorch-ROCm.
"""
import numpy as np


def lucyd_input(cleaned_u8):
    """uint8 [n,H,W] (device) -> float32 [n,1,H,W]: ``(cleaned / 255)`` evaluated in float64 and then narrowed, exactly
    what ``torch.from_numpy(cleaned / 255) ... .float()`` gives (deconvolution.py:76-78,91)."""
    import torch
    return cleaned_u8.to(torch.float64).div_(255.0).to(torch.float32).unsqueeze(1)


def lucyd_output_to_u8(y_hat):
    """float32 [n,1,H,W] -> uint8 [n,H,W]: ``(deconv * 255).astype(np.uint8)`` (deconvolution.py:96-97): truncation toward
    zero; values outside 0..255 wrap modulo 256 as NumPy's cast does on x86-64."""
    import torch
    y = y_hat[:, 0].to(torch.float32) * 255.0
    return y.to(torch.int64).to(torch.uint8)               # via int64: modular wrap, identical to the host cast


def process_with_deconvolution(seg, frames, model, mask=None, batch_size=4, n_bg_imgs=6, threshold=10,
                               min_area_to_segment=10.0, min_area_to_save=400.0, labels=None, fetch_crops=False):
    """segmenter.py:241-254 with deconvolution=True for one batch resident on the device.  ``model`` is any callable
    mapping float32 [b,1,H,W] to a tensor of the same shape or to a tuple whose first element is that (LUCYD returns
    ``(y_hat, k, update)``).  Corrupt frames (raw std <= 2) skip the network, as in deconvolution.py:73,102-103.
    Returns (bg_corrected, deconvolved, mean_std, valid, table)."""
    import torch
    cleaned, ms, valid = seg.bgcorrect(frames, n_bg_imgs=n_bg_imgs)
    deconv = torch.zeros_like(cleaned)
    good = np.nonzero(valid)[0]
    with torch.no_grad():
        for a in range(0, len(good), batch_size):
            idx = torch.as_tensor(good[a:a + batch_size], device=cleaned.device)
            y = model(lucyd_input(cleaned.index_select(0, idx)))
            if isinstance(y, (tuple, list)):
                y = y[0]
            deconv.index_copy_(0, idx, lucyd_output_to_u8(y))
    tab = seg.detect(deconv, raw=cleaned, mask=mask, valid=valid, threshold=threshold, min_area_to_segment=min_area_to_segment,
                     min_area_to_save=min_area_to_save, labels=labels, fetch_crops=fetch_crops)
    return cleaned, deconv, ms, valid, tab


def run_deconvolution(input, output, n_imgs, batch_size=4, model=None, device=0):
    """Drop-in for deconvolution.py:39 (``run_deconvolution(input, output, n_imgs, batch_size)``): takes the tuples of the
    background stage from ``input``, runs ``model`` on batches of frames with raw std > 2 and puts
    ``(bg_corrected, deconvolved, [mean, std], fn)`` on ``output`` (deconvolution.py:70-100); corrupt frames pass through
    unchanged (:102-103).  In the reference the network is a module-level global; here it is the ``model`` argument.
    Frames are kept on the device (the detection stage accepts tensors)."""
    import torch
    if model is None:
        raise ValueError("run_deconvolution needs the network as model= (the reference's LUCYD module and weights live in its repository)")
    dev = f"cuda:{device}"
    pend = []

    def flush():
        if not pend:
            return
        cl = torch.stack([p[1] for p in pend])
        with torch.no_grad():
            y = model(lucyd_input(cl))
        y = y[0] if isinstance(y, (tuple, list)) else y
        d = lucyd_output_to_u8(y)
        for k, (corr, _, mean, fn) in enumerate(pend):
            output.put((corr, d[k], mean, fn))
        pend.clear()

    while n_imgs > 0:
        corrected, cleaned, mean, fn = input.get()
        n_imgs -= 1
        if mean[1] > 2:                                                     # deconvolution.py:73
            to_dev = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            pend.append((to_dev(corrected), to_dev(cleaned), mean, fn))
            if len(pend) == batch_size or n_imgs == 0:
                flush()
        else:
            output.put((corrected, cleaned, mean, fn))                      # :102-103
    flush()
