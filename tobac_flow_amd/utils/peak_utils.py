"""Local-maximum detection used by detection.get_peak_filter (host glue).

Restates skimage.feature.peak_local_max (scikit-image 0.18: feature/peak.py:114-290 with
`_get_peak_mask`, `_exclude_border`, `_get_high_intensity_peaks`, `_shared/coord.py:ensure_spacing`)
for the call the reference makes: peak_local_max(image2d, min_distance=d) with every other argument
at its default.  scikit-image is a third-party dependency that is absent from the build image."""
import numpy as np
import scipy.ndimage as ndi


def peak_local_max(image, min_distance=1, threshold_abs=None, threshold_rel=None, exclude_border=True,
                   num_peaks=np.inf):
    image = np.asarray(image)
    threshold = image.min() if threshold_abs is None else threshold_abs
    if threshold_rel is not None:
        threshold = max(threshold, threshold_rel * image.max())
    size = 2 * min_distance + 1
    if size == 1 or image.size == 1:
        mask = image > threshold
    else:
        mask = image == ndi.maximum_filter(image, footprint=np.ones((size,) * image.ndim, bool), mode="constant")
        if np.all(mask):
            mask[:] = False
        mask &= image > threshold
    border = min_distance if exclude_border is True else int(exclude_border)
    if border:
        for ax in range(mask.ndim):
            sl = [slice(None)] * mask.ndim
            sl[ax] = slice(None, border)
            mask[tuple(sl)] = False
            sl[ax] = slice(-border, None)
            mask[tuple(sl)] = False
    coords = np.transpose(np.nonzero(mask))
    return select_peaks(coords, image[tuple(coords.T)], min_distance, num_peaks)


def select_peaks(coords, intensities, min_distance, num_peaks=np.inf):
    """Second half of peak_local_max (scikit-image 0.18 `_get_high_intensity_peaks` + `ensure_spacing`): candidates
    in C order with their intensities -> highest first, greedily dropping anything closer than min_distance."""
    coords = np.asarray(coords)
    if len(coords) == 0:
        return coords
    coords = coords[np.argsort(-np.asarray(intensities))]
    keep, kept = np.ones(len(coords), bool), []
    for i, c in enumerate(coords):            # highest first; drop anything closer than min_distance (Chebyshev)
        if kept and np.min(np.max(np.abs(np.array(kept) - c), axis=1)) < min_distance:
            keep[i] = False
        else:
            kept.append(c)
    coords = coords[keep]
    return coords[:int(num_peaks)] if len(coords) > num_peaks else coords


__all__ = ("peak_local_max", "select_peaks")
