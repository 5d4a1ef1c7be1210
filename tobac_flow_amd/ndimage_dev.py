"""Device-resident versions of the scipy.ndimage / numpy glue the detection recipes use (SURVEY.md
section 8f-2).  Every function takes and returns torch tensors on the GPU and is bit-exact with the SciPy /
numpy call it replaces (tests/test_gpu_detection.py); the numpy paths of detection.py keep using SciPy."""
import numpy as np

from tobac_flow_amd import _lib


def _structure27(structure):
    s = np.asarray(structure) != 0
    if s.shape != (3, 3, 3):
        raise ValueError("structure must be a (3, 3, 3) array")
    return np.ascontiguousarray(s, np.uint8)


def _as_u8_mask(x):
    """non-zero -> 1 as a contiguous uint8 tensor; a bool tensor is reinterpreted in place (its bytes are 0 / 1 already)"""
    t = _lib.torch()
    if x.dtype == t.bool:
        return x.contiguous().view(t.uint8)
    return (x != 0).contiguous().view(t.uint8)


def _binary(t_in, structure, op, iterations, border_value):
    t = _lib.torch()
    L = _lib.lib()
    x = _as_u8_mask(t_in)
    T, H, W = x.shape
    out = t.empty_like(x)
    tmp = t.empty_like(x) if iterations > 1 else None
    st = _structure27(structure)
    _lib.check(L.tf_binary_morph(_lib.ptr(x), T, H, W, st.ctypes.data_as(_lib._P), op, int(iterations), int(bool(border_value)),
                                 _lib.ptr(out), _lib.ptr(tmp), _lib.stream_ptr()), "tf_binary_morph")
    return out.view(t.bool)                       # the kernels write 0 / 1


def binary_erosion(x, structure, iterations=1, border_value=0):
    """scipy.ndimage.binary_erosion(x, structure=structure, iterations=iterations, border_value=border_value)"""
    return _binary(x, structure, 0, iterations, border_value)


def binary_dilation(x, structure, iterations=1, border_value=0):
    return _binary(x, structure, 1, iterations, border_value)


def binary_opening(x, structure, iterations=1):
    """scipy.ndimage.binary_opening: erosion then dilation, border_value 0"""
    return binary_dilation(binary_erosion(x, structure, iterations), structure, iterations)


def linearise_field(field, lower_threshold, upper_threshold):
    """utils.normalisation_utils.linearise_field on a float32 device tensor"""
    t = _lib.torch()
    if lower_threshold == upper_threshold:
        raise ValueError("lower and upper thresholds must have different values")
    f = field.to(t.float32).contiguous()
    out = t.empty_like(f)
    _lib.check(_lib.lib().tf_linearise(_lib.ptr(f), f.numel(), float(lower_threshold), float(upper_threshold),
                                       _lib.ptr(out), _lib.stream_ptr()), "tf_linearise")
    return out


def field_masks(field):
    """(field >= 1, (field <= 0) | isnan(field), isnan(field)) as bool tensors from ONE read of a float32 field
    (tf_field_masks: the thresholds of detect_anvils' seeds, detection.py:551-552, 607-609)"""
    t = _lib.torch()
    f = field.to(t.float32).contiguous()
    ge1, le0, isn = (t.empty(f.shape, dtype=t.uint8, device=f.device) for _ in range(3))
    _lib.check(_lib.lib().tf_field_masks(_lib.ptr(f), f.numel(), _lib.ptr(ge1), _lib.ptr(le0), _lib.ptr(isn), _lib.stream_ptr()),
               "tf_field_masks")
    return ge1.view(t.bool), le0.view(t.bool), isn.view(t.bool)


def merge_seeds(comp, bg, isnan):
    """seeds = -1 where (bg | isnan) else comp (tf_merge_seeds; detection.py:558, 616)"""
    t = _lib.torch()
    c = comp.to(t.int32).contiguous()
    out = t.empty_like(c)
    _lib.check(_lib.lib().tf_merge_seeds(_lib.ptr(c), _lib.ptr(_as_u8_mask(bg)), _lib.ptr(_as_u8_mask(isnan)), c.numel(), _lib.ptr(out),
                                         _lib.stream_ptr()), "tf_merge_seeds")
    return out


def label_extent(labels, mask=None):
    """(lengths, hit): for labels 1..max the extent along the leading axis (analysis.find_object_lengths) and whether
    the label overlaps `mask` (analysis.mask_labels); numpy arrays of length max label."""
    t = _lib.torch()
    lab = labels.to(t.int32).contiguous()
    T, H, W = lab.shape
    n = int(lab.max().item()) if lab.numel() else 0
    n = max(n, 0)
    tmin = t.empty(n + 1, dtype=t.int32, device=lab.device)
    tmax = t.empty(n + 1, dtype=t.int32, device=lab.device)
    hit = t.empty(n + 1, dtype=t.uint8, device=lab.device)
    m = None if mask is None else (mask != 0).to(t.uint8).contiguous()
    _lib.check(_lib.lib().tf_label_extent(_lib.ptr(lab), _lib.ptr(m), T, H, W, n, _lib.ptr(tmin), _lib.ptr(tmax),
                                          _lib.ptr(hit), _lib.stream_ptr()), "tf_label_extent")
    tmin, tmax = tmin[1:].cpu().numpy().astype(np.int64), tmax[1:].cpu().numpy().astype(np.int64)
    lengths = np.where(tmax >= 0, tmax - tmin + 1, 0)
    return lengths, hit[1:].cpu().numpy().astype(bool)


def remap_labels(labels, locations):
    """utils.label_utils.remap_labels(labels, locations) for a boolean `locations` of length max label"""
    t = _lib.torch()
    lab = labels.to(t.int32).contiguous()
    locations = np.asarray(locations, bool)
    lut = np.zeros(locations.size + 1, np.int32)
    lut[1:][locations] = np.arange(1, int(locations.sum()) + 1)
    lut_t = t.from_numpy(lut).to(lab.device)
    out = t.empty_like(lab)
    _lib.check(_lib.lib().tf_apply_lut(_lib.ptr(lab), lab.numel(), _lib.ptr(lut_t), lut.size, _lib.ptr(out), _lib.stream_ptr()),
               "tf_apply_lut")
    return out


def label(x, structure=None):
    """scipy.ndimage.label(x, structure) -> (int32 labels tensor, n_labels).  `structure`: (3, 3, 3), symmetric;
    default = ndi.generate_binary_structure(3, 1)."""
    import ctypes
    import scipy.ndimage as ndi
    t = _lib.torch()
    L = _lib.lib()
    xb = _as_u8_mask(x)
    T, H, W = xb.shape
    st = _structure27(ndi.generate_binary_structure(3, 1) if structure is None else structure)
    out = t.empty((T, H, W), dtype=t.int32, device=xb.device)
    ws = _lib.workspace(L.tf_label_workspace_bytes(T, H, W), "label")
    n = ctypes.c_int(0)
    _lib.check(L.tf_label(_lib.ptr(xb), T, H, W, st.ctypes.data_as(_lib._P), _lib.ptr(out), ctypes.byref(n),
                          _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "tf_label")
    return out, n.value


def flat_label(mask, structure=None):
    """utils.label_utils.flat_label: components that never connect across the leading (time) axis"""
    import scipy.ndimage as ndi
    s = np.array(ndi.generate_binary_structure(3, 1) if structure is None else structure, bool).copy()
    s[0] = 0
    s[-1] = 0
    return label(mask, s)[0]


def _float_type(x):
    t = _lib.torch()
    if x.dtype == t.float32:
        return _lib.TF_F32
    if x.dtype == t.float64:
        return _lib.TF_F64
    raise TypeError(f"float32 or float64 tensor required, got {x.dtype}")


def gaussian_kernel1d(sigma, truncate=4.0):
    """(weights, radius) of scipy.ndimage.gaussian_filter1d(order=0): radius = int(truncate * sigma + 0.5),
    exp(-0.5 / sigma^2 * x^2) normalised to sum 1, float64"""
    sd = float(sigma)
    radius = int(truncate * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return np.ascontiguousarray(phi / phi.sum(), np.float64), radius


def gaussian_filter(x, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter(x, sigma, mode='reflect', truncate=truncate) for a (T, H, W) float32 / float64
    device tensor: one symmetric correlate1d pass per axis with sigma > 1e-15, in axis order, each pass rounding to
    x's dtype (SciPy filters the later axes in place on the output of the earlier ones)."""
    t = _lib.torch()
    L = _lib.lib()
    ty = _float_type(x)
    sig = [float(v) for v in (sigma if np.ndim(sigma) else [sigma] * 3)]
    if x.dim() != 3 or len(sig) != 3:
        raise ValueError("gaussian_filter: a (T, H, W) tensor and a scalar or 3 sigmas are required")
    if any(v < 0 for v in sig):
        raise ValueError("gaussian_filter: sigma must be non-negative")
    cur = x.contiguous()
    T, H, W = cur.shape
    bufs = [None, None]
    k = 0
    for axis, sd in enumerate(sig):
        if sd <= 1e-15:
            continue
        w, r = gaussian_kernel1d(sd, truncate)
        if bufs[k] is None:
            bufs[k] = t.empty_like(cur)
        out = bufs[k]
        _lib.check(L.tf_correlate1d_sym(_lib.ptr(cur), ty, T, H, W, axis, w.ctypes.data_as(_lib._P), r, _lib.ptr(out),
                                        _lib.stream_ptr()), "tf_correlate1d_sym")
        cur, k = out, k ^ 1
    return cur.clone() if cur is x else cur


def _grey(x, footprint, op):
    t = _lib.torch()
    ty = _float_type(x)
    fp = np.asarray(footprint) != 0
    while fp.ndim < 3:
        fp = fp[np.newaxis]
    if fp.ndim != 3 or any(n not in (1, 3) for n in fp.shape):
        raise ValueError("footprint axes must have length 1 or 3")
    full = np.zeros((3, 3, 3), np.uint8)                     # centre the footprint in a 3x3x3 box
    full[tuple(slice(1, 2) if n == 1 else slice(0, 3) for n in fp.shape)] = fp
    if not np.array_equal(full, full[::-1, ::-1, ::-1]):
        raise ValueError("footprint must be point-symmetric (SciPy mirrors it for dilations)")
    if fp.all() and fp.size > 1:
        raise NotImplementedError("an all-true footprint takes SciPy's separable min/max path, whose NaN handling "
                                  "differs from the footprint path implemented here")
    xc = x.contiguous()
    T, H, W = xc.shape
    out = t.empty_like(xc)
    _lib.check(_lib.lib().tf_grey_morph(_lib.ptr(xc), ty, T, H, W, np.ascontiguousarray(full).ctypes.data_as(_lib._P), op,
                                        _lib.ptr(out), _lib.stream_ptr()), "tf_grey_morph")
    return out


def grey_erosion(x, footprint):
    """scipy.ndimage.grey_erosion(x, footprint=footprint) (flat footprint, mode 'reflect')"""
    return _grey(x, footprint, 0)


def grey_dilation(x, footprint):
    """scipy.ndimage.grey_dilation(x, footprint=footprint) (flat footprint, mode 'reflect')"""
    return _grey(x, footprint, 1)


def grey_opening(x, footprint):
    """scipy.ndimage.grey_opening(x, footprint=footprint): erosion then dilation"""
    return grey_dilation(grey_erosion(x, footprint), footprint)


def binary_fill_holes(x, structure):
    """scipy.ndimage.binary_fill_holes(x, structure): SciPy floods the background from outside the array
    (binary_dilation of the empty set, border_value 1, mask = ~x, to convergence) and returns what the flood does
    not reach.  Here: the background components (tf_label, same structure) that contain a pixel whose structure
    neighbourhood leaves the volume are 'outside'; every other background pixel is a hole."""
    t = _lib.torch()
    xb = x != 0
    bg = ~xb
    edge = binary_dilation(t.zeros_like(xb), structure, 1, border_value=1)      # one SciPy dilation step from outside
    lab, n = label(bg, structure)
    if n == 0:
        return xb.clone()
    _, outside = label_extent(lab, mask=edge & bg)
    lut = np.zeros(n + 1, np.int32)
    lut[1:][outside[:n]] = 1
    lut_t = t.from_numpy(lut).to(lab.device)
    reached = t.empty_like(lab)
    _lib.check(_lib.lib().tf_apply_lut(_lib.ptr(lab), lab.numel(), _lib.ptr(lut_t), lut.size, _lib.ptr(reached),
                                       _lib.stream_ptr()), "tf_apply_lut")
    return ~(reached != 0)


def peak_local_max_2d(image, min_distance=1):
    """skimage.feature.peak_local_max(image2d, min_distance=d) (scikit-image 0.18, every other argument at its default)
    for a 2-D float device tensor; returns the (n, 2) int64 numpy array utils.peak_utils.peak_local_max returns.
    The candidate mask (separable maximum filter, threshold = image minimum, border exclusion) is built on the GPU;
    the few candidates go to the host, where the SAME selection code as the host function orders and thins them."""
    import torch.nn.functional as F
    from tobac_flow_amd.utils.peak_utils import select_peaks
    t = _lib.torch()
    _float_type(image)
    if image.dim() != 2:
        raise ValueError("peak_local_max_2d: a 2-D tensor is required")
    d = int(min_distance)
    H, W = image.shape
    threshold = image.min()                                   # NaN anywhere -> NaN threshold -> no peak, as in numpy
    size = 2 * d + 1
    if size == 1 or image.numel() == 1:
        mask = image > threshold
    else:
        # maximum_filter(footprint = ones, mode = 'constant', cval = 0): zero padding, then max over the window
        padded = F.pad(image[None, None], (d, d, d, d), mode="constant", value=0.0)
        mx = F.max_pool2d(F.max_pool2d(padded, kernel_size=(1, size), stride=1), kernel_size=(size, 1), stride=1)[0, 0]
        mask = image == mx
        if bool(mask.all().item()):
            mask = t.zeros_like(mask)
        mask = mask & (image > threshold)
    if d:
        mask = mask.clone()
        mask[:d] = False
        mask[-d:] = False
        mask[:, :d] = False
        mask[:, -d:] = False
    coords = t.nonzero(mask)                                  # row-major, like np.nonzero
    vals = image[coords[:, 0], coords[:, 1]] if coords.shape[0] else image.new_zeros(0)
    return select_peaks(coords.cpu().numpy(), vals.cpu().numpy(), d)


_DISC_CACHE = {}


def within_distance(points, shape, radius, device):
    """scipy.ndimage.distance_transform_edt(~peaks) < radius for a 2-D frame whose only non-zero pixels are `points`
    ((n, 2) int array): squared distances are integers, so this is the union of the discs dy^2 + dx^2 < radius^2.
    A frame WITHOUT any point has no background for the transform; SciPy (1.7 and 1.15 alike) then returns
    sqrt((y + 1)^2 + x^2), i.e. the test marks a small corner at the origin -- reproduced as is."""
    t = _lib.torch()
    H, W = shape
    r2 = float(radius) ** 2
    out = t.zeros((H, W), dtype=t.bool, device=device)
    pts = np.asarray(points, np.int64).reshape(-1, 2)
    if len(pts) == 0:
        yy = t.arange(H, device=device)[:, None] + 1
        xx = t.arange(W, device=device)[None, :]
        return (yy * yy + xx * xx).to(t.float64) < r2
    key = float(radius)
    if key not in _DISC_CACHE:
        k = int(np.ceil(radius))
        dy, dx = np.mgrid[-k:k + 1, -k:k + 1]
        keep = (dy * dy + dx * dx) < r2
        _DISC_CACHE[key] = np.stack([dy[keep], dx[keep]], 1).astype(np.int64)
    offs = _DISC_CACHE[key]
    cells = (pts[:, None, :] + offs[None, :, :]).reshape(-1, 2)
    ok = (cells[:, 0] >= 0) & (cells[:, 0] < H) & (cells[:, 1] >= 0) & (cells[:, 1] < W)
    cells = t.from_numpy(cells[ok]).to(device)
    out[cells[:, 0], cells[:, 1]] = True
    return out
