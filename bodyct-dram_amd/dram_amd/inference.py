"""Whole-scan, per-lobe inference on the GPU.

Device-resident restatement of the reference's working inference semantics,
`LesionSegChunkTrain.evaluate_scan` (dram/job_runner.py:720-779): for every lobe label, crop
its bounding box (+5 mm), mask outside-lobe voxels to -2048 HU, window (-1000,-300) -> [0,1],
resample to RESAMPLE_SIZE (80^3), run the model, sigmoid, resize back to the crop size and paste
where the lobe is; then the thresholding of `LesionSegTest.run` (job_runner.py:1003-1005):
Otsu on the 8-bit heat map inside the lungs (`binary_cam`, dram/utils.py:226-242), mask = htp > th.

Differences from the reference, on purpose: the lobes (every non-zero label of the map, like
`np.unique(lobe)[1:]`; normally 5) go through the model in batches of up to 8 chunks; the scan and the label map are uploaded once and every step runs on the device
(the reference round-trips each lobe through numpy and SimpleITK).  The crop -> 80^3 resampling restates the grid of the
reference's call -- sitk.ResampleImageFilter with the identity transform, the crop's own origin and the output spacing
spacing * size_in / size_out, linear, default value 0 (utils.py:371-381) -- from ITK's published semantics: output voxel o
samples continuous input index o * size_in / size_out, zero beyond size_in - 0.5 (csrc/infer.hip:itk_index).  SimpleITK is not
available here, so that step's parity is unpinned; the way back is the reference's own F.interpolate(align_corners=True).
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import call

# IntRegLoss.ctss_ratio_map / ratio_to_label (dram/metrics.py:76-83, 108-114)
CTSS_RATIO_MAP = {0: (0.0, 0.001), 1: (0.001, 0.01), 2: (0.01, 0.05), 3: (0.05, 0.35), 4: (0.35, 0.5), 5: (0.5, 1.00001)}


def ratio_to_label(ratio):
    return [k for k, (lo, hi) in CTSS_RATIO_MAP.items() if lo <= ratio < hi][0]


def otsu_threshold_from_hist(hist):
    """skimage.filters.threshold_otsu on a uint8 image, restated from its published algorithm
    (maximise the between-class variance over the histogram of the occupied value range) and
    evaluated on the 256-bin histogram.  Returns the threshold in 0..255."""
    hist = np.asarray(hist, dtype=np.float64)
    nz = np.nonzero(hist)[0]
    lo, hi = int(nz[0]), int(nz[-1])
    h = hist[lo:hi + 1]
    centers = np.arange(lo, hi + 1, dtype=np.float64)
    w1 = np.cumsum(h)
    w2 = np.cumsum(h[::-1])[::-1]
    m1 = np.cumsum(h * centers) / w1
    m2 = (np.cumsum((h * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return float(centers[int(np.argmax(var12))])


def binary_cam_threshold(hist, scaler=1.0):
    """`binary_cam` (dram/utils.py:226-242) on the 8-bit histogram: returns th / 255."""
    hist = np.asarray(hist)
    if np.count_nonzero(hist) < 2:                      # only one colour
        return float(np.nonzero(hist)[0][0]) / 255.0
    return min(otsu_threshold_from_hist(hist) * scaler, 255.0) / 255.0


_RESAMPLE_KINDS = {torch.uint8: 0, torch.int16: 1, torch.float32: 2}


def resample_volume(vol, spacing, required_spacing, new_size, interpolator="linear"):
    """`utils.resample(narray, spacing, required_spacing=..., new_size=..., interpolator=...)` (dram/utils.py:414-434) of a
    device volume [D,H,W] (uint8 / int16 / float32; the pixel type is kept like sitk's `orig_pixelid`): the grid of
    sitk.ResampleImageFilter restated from ITK's published semantics (csrc/infer.hip: resample_volume_kernel; SimpleITK is not
    available, parity unpinned).  spacing / required_spacing / new_size in (z, y, x) order like the reference's numpy side."""
    if interpolator not in ("nearest", "linear"):
        raise NotImplementedError(f"interpolator {interpolator!r}: the reference's inference uses 'nearest' and 'linear'")
    if vol.dim() != 3 or vol.dtype not in _RESAMPLE_KINDS or not vol.is_cuda:
        raise ValueError("resample_volume: a [D,H,W] uint8 / int16 / float32 tensor on the GPU")
    vol = vol.contiguous()
    size = [int(v) for v in new_size]
    out = torch.empty(size, dtype=vol.dtype, device=vol.device)
    sp_in = (ctypes.c_double * 3)(*[float(v) for v in spacing])
    sp_out = (ctypes.c_double * 3)(*[float(v) for v in required_spacing])
    call("dram_resample_volume", vol.data_ptr(), out.data_ptr(), _RESAMPLE_KINDS[vol.dtype], int(interpolator == "linear"),
         *vol.shape, *size, sp_in, sp_out, torch.cuda.current_stream().cuda_stream)
    return out


class LobeInference:
    """model: a models.DC3D on the GPU.  run(scan_i16, lobe_u8, spacing) -> dict.  Every label 1..max_labels
    present in the uint8 lobe map is visited in ascending order (evaluate_scan: `np.unique(lobe)[1:]`,
    job_runner.py:729); a label above max_labels raises instead of being silently left out of the heat map."""

    MAX_CHUNKS = 8     # chunks per kernel call / model batch (csrc/infer.hip MAX_CHUNKS)

    def __init__(self, model, resample_size=80, window=(-1000.0, -300.0), border_mm=5.0, max_labels=255,
                 post_window=(-1150, 350), post_scaler=0.75):
        if not 1 <= int(max_labels) <= 255:
            raise ValueError("max_labels must be in 1..255 (uint8 label map)")
        self.model, self.R, self.window, self.border, self.max_labels = model, int(resample_size), window, border_mm, int(max_labels)
        # the brightness gate of LesionSegTest.run: windowing()'s default span (utils.py:189, job_runner.py:1006) and the 0.75
        # on its Otsu threshold (job_runner.py:1007)
        self.post_window, self.post_scaler = (int(post_window[0]), int(post_window[1])), float(post_scaler)

    @torch.no_grad()
    def run(self, scan, lobe, spacing, vessel=None, lesion=None, original_spacing=None, original_size=None):
        """`vessel` (uint8 [D,H,W], optional): the vessel mask of LesionSegTest.run; with it (or with `lesion`) the result also
        holds the post-processed mask `mask_post` = mask & (windowed scan > brightness threshold) & ~vessel (job_runner.py:1006-1010).
        `lesion` (uint8 [D,H,W], optional): the reference mask; adds iou / iou_post / dice / dice_post (job_runner.py:1033-1037),
        computed at the working resolution unless `original_spacing` and `original_size` ((z, y, x), the scan's metadata) are
        given: then, like job_runner.py:1016-1032, the masks (nearest neighbour), the scan and the heat map (linear) first go
        back to the original grid (`resample_volume`; result key "original": mask / mask_post / lesion / scan / htp there) and
        the four metrics are those of the resampled masks."""
        dev = next(self.model.parameters()).device
        scan = torch.as_tensor(scan).to(device=dev, dtype=torch.int16).contiguous()
        lobe = torch.as_tensor(lobe).to(device=dev, dtype=torch.uint8).contiguous()
        if scan.dim() != 3 or scan.shape != lobe.shape:
            raise ValueError("scan and lobe must be [D,H,W] arrays of the same shape")
        D, H, W = scan.shape
        st = torch.cuda.current_stream().cuda_stream
        boxes = torch.empty(self.max_labels * 6, dtype=torch.int32, device=dev)
        call("dram_label_bboxes", lobe.data_ptr(), boxes.data_ptr(), self.max_labels, D, H, W, st)
        top = int(lobe.max()) if self.max_labels < 255 else 0             # (a second small sync only when labels are capped)
        if top > self.max_labels:
            raise ValueError(f"lobe map holds label {top} > max_labels={self.max_labels}: it would be left out of the "
                             f"heat map while still counting in the lesion ratio")
        boxes = boxes.cpu().numpy().reshape(self.max_labels, 6)            # the only sync before the model: 6 ints per label
        chunks = []
        for label in range(1, self.max_labels + 1):
            lo, hi = boxes[label - 1, :3], boxes[label - 1, 3:]
            if hi[0] < lo[0]:
                continue                                                     # label absent (np.unique(lobe)[1:])
            start, stop = [], []
            for ax, (l, h, size, sp) in enumerate(zip(lo, hi, (D, H, W), spacing)):   # find_crops, utils.py:244-254
                pad = int(math.ceil(self.border / sp)) if self.border > 0 else 0
                start.append(max(0, int(l) - pad))
                stop.append(min(size, int(h) + 1 + pad))
            chunks.append(start + [b - a for a, b in zip(start, stop)] + [label])
        htp = torch.zeros((D, H, W), dtype=torch.float32, device=dev)
        if not chunks:
            return {"htp": htp, "mask": torch.zeros((D, H, W), dtype=torch.uint8, device=dev), "threshold": 0.0,
                    "lesion_ratio": 0.0, "ctss": 0, "chunks": []}
        R = self.R
        x = torch.empty((len(chunks), 1, R, R, R), dtype=torch.float32, device=dev)
        was_training = self.model.training
        self.model.eval()
        try:
            for g0 in range(0, len(chunks), self.MAX_CHUNKS):
                grp = chunks[g0:g0 + self.MAX_CHUNKS]
                L = len(grp)
                carr = (ctypes.c_int * (7 * L))(*[v for c in grp for v in c])
                xg = x[g0:g0 + L]
                call("dram_lobe_chunks", scan.data_ptr(), lobe.data_ptr(), xg.data_ptr(), carr, L, D, H, W, R,
                     float(self.window[0]), float(self.window[1]), st)
                _, dense = self.model(xg, None)   # one batch of L lobe chunks; the 2nd output is used (job_runner.py:764)
                call("dram_lobe_paste", dense.contiguous().data_ptr(), lobe.data_ptr(), htp.data_ptr(), carr, L, D, H, W, R, st)
        finally:
            self.model.train(was_training)
        hist = torch.empty(256, dtype=torch.int64, device=dev)
        ssum = torch.empty(1, dtype=torch.float64, device=dev)
        call("dram_lung_hist256", htp.data_ptr(), lobe.data_ptr(), hist.data_ptr(), ssum.data_ptr(), htp.numel(), st)
        hist_h = hist.cpu().numpy()
        n_lung = int(hist_h.sum())
        th = binary_cam_threshold(hist_h)
        mask = torch.empty((D, H, W), dtype=torch.uint8, device=dev)
        call("dram_threshold_mask", htp.data_ptr(), mask.data_ptr(), float(th), htp.numel(), st)
        ratio = float(ssum.item()) / max(n_lung, 1)                          # (htp * (lobe>0)).sum() / (lobe>0).sum()
        res = {"htp": htp, "mask": mask, "threshold": th, "lesion_ratio": ratio, "ctss": ratio_to_label(ratio),
               "chunks": chunks, "input": x}
        if vessel is not None or lesion is not None or original_size is not None:
            res.update(self.post_process(scan, lobe, htp, th, vessel, lesion, mask, spacing=spacing,
                                         original_spacing=original_spacing, original_size=original_size))
        return res

    @torch.no_grad()
    def post_process(self, scan, lobe, htp, th, vessel=None, lesion=None, mask=None, spacing=None, original_spacing=None,
                     original_size=None):
        """The tail of LesionSegTest.run (job_runner.py:1006-1037) on the device: brightness gate
        `binary_cam(w_scan[lobe > 0], 0.75)` on the scan windowed with windowing()'s default span, lesion_pred_post =
        lesion_pred & (w_scan > th) & ~(vessel > 0), the resampling to the scan's original grid when `original_spacing` /
        `original_size` are given (job_runner.py:1016-1032), and IOU / Dice against the reference lesion mask (utils.py:437-446)."""
        if (original_spacing is None) != (original_size is None) or (original_size is not None and spacing is None):
            raise ValueError("post_process: spacing, original_spacing and original_size go together")
        dev = htp.device
        st = torch.cuda.current_stream().cuda_stream
        n = htp.numel()
        vessel = None if vessel is None else torch.as_tensor(vessel).to(device=dev, dtype=torch.uint8).contiguous()
        if vessel is not None and vessel.shape != htp.shape:
            raise ValueError("vessel mask and scan must have the same shape")
        hist = torch.empty(256, dtype=torch.int64, device=dev)
        call("dram_scan_hist256", scan.data_ptr(), lobe.data_ptr(), hist.data_ptr(), self.post_window[0], self.post_window[1], n, st)
        th_scan = binary_cam_threshold(hist.cpu().numpy(), self.post_scaler)
        post = torch.empty(htp.shape, dtype=torch.uint8, device=dev)
        call("dram_lesion_post", htp.data_ptr(), scan.data_ptr(), None if vessel is None else vessel.data_ptr(), None,
             post.data_ptr(), float(th), self.post_window[0], self.post_window[1], float(th_scan), n, st)
        out = {"mask_post": post, "threshold_scan": th_scan}
        if mask is None and (lesion is not None or original_size is not None):
            mask = torch.empty(htp.shape, dtype=torch.uint8, device=dev)
            call("dram_threshold_mask", htp.data_ptr(), mask.data_ptr(), float(th), n, st)
        if lesion is not None:
            lesion = torch.as_tensor(lesion).to(device=dev, dtype=torch.uint8).contiguous()
            if lesion.shape != htp.shape:
                raise ValueError("lesion mask and scan must have the same shape")
        if original_size is not None:
            back = lambda v, how: resample_volume(v, spacing, original_spacing, original_size, how)
            orig = {"mask": back(mask, "nearest"), "mask_post": back(post, "nearest"), "scan": back(scan, "linear"),
                    "htp": back(htp, "linear")}
            if lesion is not None:
                orig["lesion"] = lesion = back(lesion, "nearest")
            out["original"] = orig
            mask, post, n = orig["mask"], orig["mask_post"], orig["mask"].numel()
        if lesion is not None:
            counts = torch.empty(8, dtype=torch.int64, device=dev)
            call("dram_mask_overlap", mask.data_ptr(), lesion.data_ptr(), counts.data_ptr(), n, st)
            call("dram_mask_overlap", post.data_ptr(), lesion.data_ptr(), counts[4:].data_ptr(), n, st)
            c = [int(v) for v in counts.cpu()]
            s = 1e-5                                                        # the smooth term of job_runner.py:1033-1037
            out.update(iou=(c[0] + s) / (c[1] + s), dice=(2.0 * c[0] + s) / (c[2] + c[3] + s),
                       iou_post=(c[4] + s) / (c[5] + s), dice_post=(2.0 * c[4] + s) / (c[6] + c[7] + s))
        return out


def iou(predict, target, smooth=1e-5):
    """dram/utils.py:437-442."""
    predict, target = np.asarray(predict) > 0, np.asarray(target) > 0
    return (np.logical_and(predict, target).sum() + smooth) / (np.logical_or(predict, target).sum() + smooth)


def dice(predict, target, smooth=1e-5):
    """dram/utils.py:444-446."""
    predict, target = np.asarray(predict) > 0, np.asarray(target) > 0
    inter = np.logical_and(predict, target).sum()
    return (2.0 * inter + smooth) / (predict.sum() + target.sum() + smooth)


def synthetic_ct(shape=(300, 512, 512), spacing=(1.0, 0.7, 0.7), seed=7, n_lesions=20):
    """air -1000, lung parenchyma N(-850,60^2) inside 5 disjoint
    ellipsoid "lobes" labelled 1-5, spherical lesions N(-300,80^2) r in [5,20] voxels, body +40."""
    rng = np.random.default_rng(seed)
    D, H, W = shape
    scan = np.full(shape, 40, dtype=np.int16)
    lobe = np.zeros(shape, dtype=np.uint8)
    zz, yy, xx = np.meshgrid(np.arange(D, dtype=np.float32), np.arange(H, dtype=np.float32),
                             np.arange(W, dtype=np.float32), indexing="ij", sparse=True)
    centres = [(0.30, 0.45, 0.27), (0.62, 0.50, 0.27), (0.28, 0.45, 0.73), (0.52, 0.36, 0.73), (0.74, 0.60, 0.73)]
    radii = [(0.17, 0.22, 0.13), (0.17, 0.24, 0.13), (0.13, 0.20, 0.12), (0.10, 0.14, 0.10), (0.13, 0.20, 0.12)]
    for lab, (c, r) in enumerate(zip(centres, radii), start=1):
        m = (((zz - c[0] * D) / (r[0] * D)) ** 2 + ((yy - c[1] * H) / (r[1] * H)) ** 2 + ((xx - c[2] * W) / (r[2] * W)) ** 2) < 1.0
        m &= lobe == 0
        lobe[m] = lab
    lung = lobe > 0
    scan[lung] = rng.normal(-850, 60, size=int(lung.sum())).astype(np.int16)
    idx = np.argwhere(lung)
    for _ in range(n_lesions):
        cz, cy, cx = idx[rng.integers(len(idx))]
        rad = rng.integers(5, 21) * min(1.0, min(shape) / 300.0)
        z0, z1 = max(0, int(cz - rad)), min(D, int(cz + rad) + 1)
        y0, y1 = max(0, int(cy - rad)), min(H, int(cy + rad) + 1)
        x0, x1 = max(0, int(cx - rad)), min(W, int(cx + rad) + 1)
        sub = ((zz[z0:z1] - cz) ** 2 + (yy[:, y0:y1] - cy) ** 2 + (xx[:, :, x0:x1] - cx) ** 2) < rad ** 2
        sub &= lung[z0:z1, y0:y1, x0:x1]
        scan[z0:z1, y0:y1, x0:x1][sub] = rng.normal(-300, 80, size=int(sub.sum())).astype(np.int16)
    return scan, lobe, spacing
