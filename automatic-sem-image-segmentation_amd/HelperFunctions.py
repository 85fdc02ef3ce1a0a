"""Host-side image helpers with the reference's names and semantics (Releases/Version 1.2.0/HelperFunctions.py):
``get_image_file_paths_from_directory`` :290-291, ``load_and_preprocess_images`` :294-329,
``tile_image`` :17-62, ``stitch_image`` :65-141.  Pure numpy/PIL; feeds and drains the GPU path."""
import math
import random
import os

import numpy as np

_EXT = ('.tif', '.tiff', '.png', '.bmp', '.jpg', '.jpeg', '.gif')


def get_image_file_paths_from_directory(directory, missing_ok=False):
    if missing_ok and not os.path.isdir(directory):
        return []
    return [os.path.join(directory, f) for f in os.listdir(directory) if f.endswith(_EXT)]


def load_and_preprocess_images(input_dir_or_filelist, threshold_value=None, normalization_range=(-1, 1),
                               output_channels=1, contrast_optimization_range=None):
    from PIL import Image
    if isinstance(input_dir_or_filelist, (str, os.PathLike)):
        files = (get_image_file_paths_from_directory(input_dir_or_filelist) if os.path.isdir(input_dir_or_filelist)
                 else [input_dir_or_filelist])
    else:
        files = input_dir_or_filelist
    images = []
    for file in files:
        image = np.array(Image.open(file), dtype='float32')
        assert 2 <= image.ndim <= 3 and output_channels in (1, 3), 'Invalid Image format'
        if image.ndim == 3 and output_channels == 1:
            image = np.average(image, -1)       # 2-D after averaging, exactly like the reference
        elif image.ndim == 2:
            image = image[:, :, np.newaxis]
        cr = contrast_optimization_range
        if cr is not None and cr[0] > 0 and cr[1] < 100:
            lb, ub = np.percentile(image, cr[0]), np.percentile(image, cr[1])
            image = np.where(image <= lb, lb, image)
            image = np.where(image >= ub, ub, image)
        if normalization_range is not None:
            image -= np.min(image)
            image /= np.max(image)
            if threshold_value is not None:
                image = image > threshold_value
            image = normalization_range[0] + (normalization_range[1] - normalization_range[0]) * image
        images.append(image)
    return np.array(images, dtype='float32')


def _tile_grid(size, tile, min_overlap):
    n = math.ceil(size / tile)
    if n > 1 and (tile - (size % tile)) % tile <= min_overlap:
        n += 1
    offs = [math.ceil(i * (tile - ((tile * n - size) / (n - 1)))) if n > 1 else 0 for i in range(n)]
    return n, offs


def tile_image(img, tile_size_w, tile_size_h, min_overlap=2, normalization_range=None, normalize_tiles_individually=True):
    h, w = img.shape[0], img.shape[1]
    nx, xs = _tile_grid(w, tile_size_w, min_overlap)
    ny, ys = _tile_grid(h, tile_size_h, min_overlap)
    tiles = np.zeros((nx * ny, tile_size_h, tile_size_w, 1), dtype='float32')
    k = 0
    for ox in xs:
        for oy in ys:
            patch = img[oy:min(oy + tile_size_h, h), ox:min(ox + tile_size_w, w), :]
            tiles[k, :, :, :] = patch
            k += 1
    if normalization_range is not None:
        lo, hi = normalization_range
        if normalize_tiles_individually:
            for i in range(tiles.shape[0]):
                tiles[i] -= np.min(tiles[i])
                tiles[i] /= np.max(tiles[i])
                tiles[i] = lo + (hi - lo) * tiles[i]
        else:
            tiles -= np.min(img)
            tiles /= np.max(img)
            tiles = lo + (hi - lo) * tiles
    return tiles


def stitch_image(img, image_size_w, image_size_h, min_overlap=2, manage_overlap_mode=2, return_8_bit_image=False):
    th, tw = img.shape[1], img.shape[2]
    nx, xs = _tile_grid(image_size_w, tw, min_overlap)
    ny, ys = _tile_grid(image_size_h, th, min_overlap)
    out = np.zeros((image_size_h, image_size_w, img.shape[-1]), dtype='float32')
    counts = np.zeros_like(out, dtype='uint8')
    ovx = (tw * nx - image_size_w) // (2 * (nx - 1)) if nx > 1 else 0
    ovy = (th * ny - image_size_h) // (2 * (ny - 1)) if ny > 1 else 0
    k = 0
    for i, ox in enumerate(xs):
        for j, oy in enumerate(ys):
            y1, x1 = min(oy + th, image_size_h), min(ox + tw, image_size_w)
            if manage_overlap_mode == 0:
                out[oy:y1, ox:x1, :] = np.maximum(img[k], out[oy:y1, ox:x1, :])
            elif manage_overlap_mode == 1:
                out[oy:y1, ox:x1, :] += img[k]
                counts[oy:y1, ox:x1, :] += 1
            elif manage_overlap_mode == 2:
                cxl = 0 if i == 0 else ovx
                cxr = 0 if i == nx - 1 else ovx
                cyt = 0 if j == 0 else ovy
                cyb = 0 if j == ny - 1 else ovy
                out[oy + cyt:min(oy + th - cyb, image_size_h), ox + cxl:min(ox + tw - cxr, image_size_w), :] = \
                    img[k, cyt:th - cyb, cxl:tw - cxr, :]
            k += 1
    if manage_overlap_mode == 1:
        out /= counts
    if return_8_bit_image:
        out = (out * 255).astype('uint8')
    return out


def threshold_otsu(image):
    """Otsu threshold of an integer image as skimage.filters.threshold_otsu computes it for uint8 data (the call at
    Measurements.py:277): histogram over the integer values min..max, threshold = value maximising the inter-class variance."""
    img = np.asarray(image)
    lo, hi = int(img.min()), int(img.max())
    if lo == hi:
        return float(lo)
    counts = np.bincount(img.ravel().astype(np.int64) - lo, minlength=hi - lo + 1).astype(np.float64)
    centers = np.arange(lo, hi + 1, dtype=np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / np.maximum(w1, 1e-300)
    m2 = (np.cumsum((counts * centers)[::-1]) / np.maximum(w2[::-1], 1e-300))[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return float(centers[int(np.argmax(var12))])


def threshold_li(image, tolerance=None):
    """Li's iterative minimum cross-entropy threshold as skimage.filters.threshold_li (0.18) computes it -- the default
    ``threshold_method`` of ``filter_gan_masks`` (HelperFunctions.py:163).  Pinned against scikit-image 0.18.3
    (tests/golden/make_postproc_goldens.py)."""
    image = np.asarray(image)
    image = image[~np.isnan(image)] if image.dtype.kind == 'f' else image.ravel()
    if image.size == 0:
        return np.nan
    if np.all(image == image.flat[0]):
        return image.flat[0]
    if np.any(np.isinf(image)):
        return np.nan
    image_min = np.min(image)
    image = image - image_min
    tolerance = tolerance or np.min(np.diff(np.unique(image))) / 2
    t_next = np.mean(image)
    t_curr = -2 * tolerance
    if image.dtype.kind in 'iu':                 # integer images: the same iteration on the histogram
        hist = np.bincount(image.ravel().astype(np.int64)).astype(float)
        centers = np.arange(hist.size, dtype=float)
        while abs(t_next - t_curr) > tolerance:
            t_curr = t_next
            fg = centers > t_curr
            bg = ~fg
            mean_fore = np.average(centers[fg], weights=hist[fg])
            mean_back = np.average(centers[bg], weights=hist[bg])
            t_next = (mean_back - mean_fore) / (np.log(mean_back) - np.log(mean_fore)) if mean_back != 0 else t_curr
    else:
        while abs(t_next - t_curr) > tolerance:
            t_curr = t_next
            fg = image > t_curr
            mean_fore = np.mean(image[fg])
            mean_back = np.mean(image[~fg])
            t_next = (mean_back - mean_fore) / (np.log(mean_back) - np.log(mean_fore)) if mean_back != 0 else t_curr
    return t_next + image_min


_POST = None


def _post_lib():
    """libsemseg_post.so (csrc/postproc.c; C ABI in include/semseg_post.h).  No Python fallback: a missing build is an error."""
    global _POST
    if _POST is None:
        import ctypes
        # SS_POST_LIB: diagnostic builds only (the AddressSanitizer build of csrc/Makefile `asan`)
        path = os.environ.get("SS_POST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsemseg_post.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C {os.path.dirname(path)}/csrc` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.ss_post_version.restype = ctypes.c_int
        lib.ss_post_watershed.restype = ctypes.c_int
        lib.ss_post_watershed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_void_p]
        lib.ss_post_eight_to_four.restype = ctypes.c_int
        lib.ss_post_eight_to_four.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        _POST = lib
    return _POST


def eight_to_four_connected(img):
    """HelperFunctions.py:131-152: break diagonal-only (8-connected) contacts, scanning in the reference's order.
    uint8 2-D images run in C (ss_post_eight_to_four), modified in place like the reference; other dtypes use the loop."""
    if isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.ndim == 2 and img.flags.c_contiguous and img.flags.writeable:
        if _post_lib().ss_post_eight_to_four(img.ctypes.data, img.shape[0], img.shape[1]) != 0:
            raise RuntimeError("ss_post_eight_to_four failed")
        return img
    if np.count_nonzero(img) > 2 or np.count_nonzero(img) < img.size - 2:
        for x in range(0, img.shape[0] - 1):
            for y in range(0, img.shape[1] - 1):
                if img[x, y] == 0 and img[x + 1, y + 1] == 0 and img[x + 1, y] != 0 and img[x, y + 1] != 0:
                    img[x + 1, y] = 0
                elif img[x + 1, y] == 0 and img[x, y + 1] == 0 and img[x, y] != 0 and img[x + 1, y + 1] != 0:
                    img[x, y] = 0
    return img


def peak_local_max(image, min_distance=1):
    """Coordinates (row, col) of local maxima separated by at least ``min_distance``, highest first -- the behaviour of
    skimage.feature.peak_local_max (0.18) with its defaults, as called at Measurements.py:290: a pixel is a candidate when
    it equals the maximum of its (2d+1)x(2d+1) window (zero outside the image) and exceeds image.min(); candidates within
    ``min_distance`` of the border are dropped; then, in order of decreasing value, a candidate suppresses every later
    candidate closer than ``min_distance`` in the Chebyshev metric.  Equal-valued candidates keep raster order (stable sort)."""
    from scipy import ndimage
    image = np.asarray(image)
    d = int(min_distance)
    size = 2 * d + 1
    cand = image == ndimage.maximum_filter(image, size=size, mode='constant')
    if np.all(cand):                      # constant image: no peaks
        cand[:] = False
    cand &= image > image.min()
    if d > 0:
        cand[:d, :] = False
        cand[-d:, :] = False
        cand[:, :d] = False
        cand[:, -d:] = False
    coords = np.transpose(np.nonzero(cand))
    if len(coords) == 0:
        return coords.astype(np.int64)
    coords = coords[np.argsort(-image[tuple(coords.T)], kind='stable')]
    # greedy spacing on a coarse grid of cell size d (a kept peak can only conflict with kept peaks in the 3x3 cells around it)
    cell = max(d, 1)
    grid = {}
    keep = []
    for r, c in coords:
        gr, gc = r // cell, c // cell
        ok = True
        for a in (gr - 1, gr, gr + 1):
            for b in (gc - 1, gc, gc + 1):
                for (pr, pc) in grid.get((a, b), ()):
                    if max(abs(pr - r), abs(pc - c)) < d:
                        ok = False
                        break
                if not ok:
                    break
            if not ok:
                break
        if ok:
            grid.setdefault((gr, gc), []).append((r, c))
            keep.append((r, c))
    return np.asarray(keep, dtype=np.int64).reshape(-1, 2)


def watershed(image, markers, mask=None, watershed_line=False):
    """skimage.segmentation.watershed(image, markers, connectivity=np.ones((3, 3)), mask=mask, watershed_line=...) for 2-D
    input, computed by ss_post_watershed (priority flooding; csrc/postproc.c)."""
    img = np.ascontiguousarray(image, dtype=np.float64)
    mk = np.ascontiguousarray(markers, dtype=np.int32)
    if img.ndim != 2 or mk.shape != img.shape:
        raise ValueError("watershed expects a 2-D image and markers of the same shape")
    mptr = None
    if mask is not None:
        mk8 = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if mk8.shape != img.shape:
            raise ValueError("mask shape differs from image shape")
        mk = np.where(mk8 != 0, mk, 0).astype(np.int32)        # markers outside the mask are ignored
        mptr = mk8.ctypes.data
    out = np.zeros(img.shape, np.int32)
    if _post_lib().ss_post_watershed(img.ctypes.data, mk.ctypes.data, mptr, img.shape[0], img.shape[1], int(bool(watershed_line)),
                                     out.ctypes.data) != 0:
        raise RuntimeError("ss_post_watershed failed")
    return out


def segment_measure(image, threshold=-1.0, applyWatershed=True, min_distance=9, darkBackground=False):
    """Measurements.Measure.segment (Measurements.py:263-305): threshold (Otsu when < 0), then split touching particles:
    Euclidean distance map of the mask, gaussian sigma=1, local maxima at least ``min_distance`` apart as markers,
    watershed of the negated map inside the mask with watershed lines.  Returns uint8 {0, 255}."""
    from scipy import ndimage
    img = np.asarray(image).copy()
    if threshold < 0:
        threshold = threshold_otsu(img)
    mask = img > threshold if darkBackground else img < threshold
    if not applyWatershed or np.min(mask) == np.max(mask):
        return np.asarray(mask * 255, dtype='uint8')
    distance = ndimage.gaussian_filter(ndimage.distance_transform_edt(mask), sigma=1)
    local_max = peak_local_max(distance, min_distance=min_distance)
    local_maxi = np.zeros(img.shape, dtype='uint8')
    local_maxi[tuple(local_max.T)] = 1
    markers = ndimage.label(local_maxi)[0]
    labels = watershed(-distance, markers, mask=mask, watershed_line=applyWatershed)
    return np.asarray((labels > 0) * 255, dtype='uint8')


def segment(image, threshold, watershed_lines, min_distance=9, use_four_connectivity=True):
    """HelperFunctions.py:155-160: Measure.segment(darkBackground=True) then optional 8->4 connectivity."""
    labels = segment_measure(image, threshold, watershed_lines, min_distance, darkBackground=True)
    if use_four_connectivity:
        labels = eight_to_four_connected(labels)
    return labels


# ---- contours as Measurements.Measure sees them (OpenCV semantics restated on pixel sets) -----------------------------------------
# `Measure.__calculateContours` (Measurements.py:158-191) calls cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE): EVERY border of
# the 8-connected foreground becomes a contour -- the outer border of each component and the border of each hole (a 4-connected
# background component that does not reach the image edge; its contour runs over the foreground pixels 4-adjacent to it, Suzuki &
# Abe 1985, the algorithm behind findContours).  What the workflow then does with a contour (HelperFunctions.py:169-178) only needs
# three pixel sets, which are restated here without tracing polygons (OpenCV is not installed in this image; pinned by hand-built
# cases in tests/test_scoring_cpu.py, not by cv2 outputs):
#   border   the contour's own pixels (drawn by cv2.drawContours whatever the fill rule does);
#   region   the integer points with cv2.pointPolygonTest(contour, p) >= 0, i.e. inside or on the polygon through the border pixels'
#            centres: for an outer border the component with its holes filled, for a hole border the hole, its ring of border pixels
#            and whatever lies inside the hole (Measurements.py:333-336 averages the grey image over exactly these points);
#   removal  contours with fewer than 5 vertices after CHAIN_APPROX_SIMPLE AND a polygon perimeter below 8 are dropped
#            (Measurements.py:176-187): only shapes inside a 4 x 4 box can qualify, those are traced explicitly (_small_polygon).
# cv2.drawContours(contourIdx=-1, thickness=-1) fills the polygons of ALL kept contours together with the even-odd rule and draws
# their borders: a hole stays a hole only if its own contour survived the filter (a dark hole does not: it gets filled).

_N8 = ((0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1))          # (dy, dx), clockwise from east (y down)


def _small_polygon(region):
    """Vertices and perimeter of the closed 8-connected outer border of a small boolean array, straight runs collapsed to their end
    points (what CHAIN_APPROX_SIMPLE keeps).  Moore-neighbour tracing; degenerate shapes as OpenCV returns them: one pixel -> one
    point, a straight line -> its two end points."""
    pts = np.argwhere(region)
    if len(pts) == 1:
        return 1, 0.0
    h, w = region.shape
    inside = lambda y, x: 0 <= y < h and 0 <= x < w and region[y, x]
    start = tuple(pts[0])                         # first pixel in raster order: nothing above it, nothing to its left in its row
    chain, cur, back = [start], start, 4          # we "came from" the west
    for _ in range(4 * region.size + 8):
        for k in range(1, 9):
            d = (back + k) % 8
            ny, nx = cur[0] + _N8[d][0], cur[1] + _N8[d][1]
            if inside(ny, nx):
                cur, back = (ny, nx), (d + 4) % 8
                break
        if cur == start and len(chain) > 1:
            # closed when we are back at the start (Jacob's criterion is not needed for shapes inside a 4 x 4 box traced from
            # their raster-first pixel: the start pixel is entered again only at the end of the walk or at a one-pixel-wide neck,
            # where the walk continues on the other side)
            nxt = None
            for k in range(1, 9):
                d = (back + k) % 8
                ny, nx = cur[0] + _N8[d][0], cur[1] + _N8[d][1]
                if inside(ny, nx):
                    nxt = (ny, nx)
                    break
            if nxt == chain[1]:
                break
        chain.append(cur)
    if chain[-1] == start and len(chain) > 1:
        chain = chain[:-1]
    n = len(chain)
    verts = [chain[i] for i in range(n)
             if (chain[i][0] - chain[i - 1][0], chain[i][1] - chain[i - 1][1]) != (chain[(i + 1) % n][0] - chain[i][0], chain[(i + 1) % n][1] - chain[i][1])]
    if not verts:
        verts = chain
    perim = sum(math.hypot(verts[i][0] - verts[i - 1][0], verts[i][1] - verts[i - 1][1]) for i in range(len(verts)))
    return len(verts), perim


def find_contours(mask):
    """The contours ``Measure`` keeps for a binary mask (excludeEdges=False), as pixel sets: a list of dicts with ``kind`` ('outer' |
    'hole'), ``slice`` (the bounding box in the image) and boolean arrays ``border`` / ``region`` over that box (see above)."""
    from scipy import ndimage
    fg = np.asarray(mask) > 0
    H, W = fg.shape
    four = ndimage.generate_binary_structure(2, 1)
    out = []

    def keep(region, sl, border):
        # Measurements.py:170-187: the small-contour test sits in the `elif` of the edge test -- a contour with a vertex on the first
        # / last row or column is never tested for smallness (and, with excludeEdges=False, always kept).  The extreme coordinates
        # of a polygon are vertices, so "a vertex on the image edge" = "a border pixel on the image edge".
        ys, xs = np.nonzero(border)
        if ys.size and (ys.min() + sl[0].start == 0 or xs.min() + sl[1].start == 0
                        or ys.max() + sl[0].start >= H - 1 or xs.max() + sl[1].start >= W - 1):
            return True
        if region.shape[0] > 4 or region.shape[1] > 4:
            return True
        nv, perim = _small_polygon(region)
        return not (nv < 5 and perim < 8)

    lab8, _ = ndimage.label(fg, structure=np.ones((3, 3)))
    for i, sl in enumerate(ndimage.find_objects(lab8), start=1):
        comp = lab8[sl] == i
        region = ndimage.binary_fill_holes(comp)
        outside = np.pad(~region, 1, constant_values=True)
        border = comp & ndimage.binary_dilation(outside, structure=four)[1:-1, 1:-1]
        if keep(region, sl, border):
            out.append(dict(kind='outer', slice=sl, border=border, region=region))
    bg4, _ = ndimage.label(~fg)                                   # 4-connected background
    edge = set(np.unique(np.concatenate([bg4[0], bg4[-1], bg4[:, 0], bg4[:, -1]]))) - {0}
    for h, sl in enumerate(ndimage.find_objects(bg4), start=1):
        if h in edge:
            continue
        big = (slice(max(sl[0].start - 1, 0), min(sl[0].stop + 1, H)), slice(max(sl[1].start - 1, 0), min(sl[1].stop + 1, W)))
        hole = bg4[big] == h
        ring = ndimage.binary_dilation(hole, structure=four) & fg[big]
        region = ndimage.binary_fill_holes(hole | ring)
        if keep(region, big, ring):
            out.append(dict(kind='hole', slice=big, border=ring, region=region))
    return out


def contour_mean_intensities(contours, gray):
    """Measure.calculateMeanIntensities (Measurements.py:321-342): sum of the grey values over the contour's region / its point count;
    0.0 when the sum is 0."""
    g = np.asarray(gray, dtype=np.float64)
    out = []
    for c in contours:
        vals = g[c['slice']][c['region']]
        tot = float(vals.sum())
        out.append(tot / vals.size if tot > 0 else 0.0)
    return out


def draw_contours_filled(contours, shape):
    """cv2.drawContours(zeros, contours, -1, 255, thickness=-1) (HelperFunctions.py:177-178): borders + even-odd fill of all the
    polygons together."""
    count = np.zeros(shape, np.int32)
    border = np.zeros(shape, bool)
    for c in contours:
        count[c['slice']] += (c['region'] & ~c['border'])
        border[c['slice']] |= c['border']
    return (((count % 2) == 1) | border).astype('uint8') * 255


def particles(mask):
    """Label image of the particles (8-connected components, holes filled) and their count -- the OUTER contours of find_contours."""
    from scipy import ndimage
    m = np.asarray(mask) > 0
    lab, n = ndimage.label(m, structure=np.ones((3, 3)))
    out = np.zeros(lab.shape, np.int32)
    for i, sl in enumerate(ndimage.find_objects(lab), start=1):
        if sl is None:
            continue
        comp = ndimage.binary_fill_holes(lab[sl] == i)
        view = out[sl]
        view[comp & (view == 0)] = i
    return out, n


def _filter_one(job):
    """One image / mask pair of ``filter_gan_masks`` (a top-level function: it also runs in worker processes)."""
    from PIL import Image, ImageFilter
    f, img_path, msk_path, out_path, threshold_method, do_watershed_and_four_connectivity, gaussian_blur_amount, dark_background = job
    img = np.array(Image.open(os.path.join(img_path, f)), dtype='uint8')
    mask = np.array(Image.open(os.path.join(msk_path, f)), dtype='uint8')
    if do_watershed_and_four_connectivity:
        mask = segment(image=mask, threshold=-1, watershed_lines=True, use_four_connectivity=True)
    contours = find_contours(mask)
    thr = threshold_method(img)
    means = contour_mean_intensities(contours, img)
    if dark_background:
        if thr != 0:          # filterResults(minValue=thr): "minValue == 0 and maxValue < minValue" returns without filtering
            contours = [c for c, m in zip(contours, means) if not m < thr]
    else:                      # filterResults(maxValue=thr): removed when mean > maxValue and maxValue >= 0; a negative maxValue filters nothing
        if thr >= 0:
            contours = [c for c, m in zip(contours, means) if not m > thr]
    Image.fromarray(draw_contours_filled(contours, img.shape)).save(os.path.join(out_path, f))
    if gaussian_blur_amount > 0:
        Image.fromarray(img).filter(ImageFilter.GaussianBlur(gaussian_blur_amount)).save(os.path.join(img_path, f))


def filter_gan_masks(img_path, msk_path, out_path, threshold_method=threshold_li, do_watershed_and_four_connectivity=True,
                     gaussian_blur_amount=0.0, dark_background=True, workers=None):
    """Workflow step 5 (StartProcess.py:133-146; HelperFunctions.py:163-185): drop simulated particles the CycleGAN did not render.
    For every generated image / mask pair: optionally re-segment the mask (Otsu, watershed lines, 4-connectivity), take the contours
    of the mask as ``Measure`` does, each contour's MEAN INTENSITY in the generated image (Measure.calculateMeanIntensities,
    Measurements.py:321-342: over the points with pointPolygonTest >= 0) and keep the contours whose mean is >= (dark background;
    <= otherwise) ``threshold_method(image)`` (Measure.filterResults('meanIntensity'), Measurements.py:569-611, including its
    "minValue == 0 and no maxValue: keep everything" shortcut); the kept contours are drawn filled (255) under the same file name;
    optional Gaussian blur of the image in place (PIL radius = ``gaussian_blur_amount``).
    The pairs are independent and draw nothing: with more than a handful they go to ``workers`` processes (SS_FILTER_WORKERS, default
    ``default_workers()``; 1 = inline) -- 1 000 pairs cost ~30 s of contour work on one core."""
    os.makedirs(out_path, exist_ok=True)
    files = sorted(os.listdir(img_path))
    jobs = [(f, img_path, msk_path, out_path, threshold_method, do_watershed_and_four_connectivity, gaussian_blur_amount, dark_background)
            for f in files]
    if workers is None:
        workers = int(os.environ.get("SS_FILTER_WORKERS", default_workers()))
    if workers > 1 and len(jobs) >= 16:
        import pickle
        try:
            pickle.dumps(threshold_method)
        except Exception:          # a lambda / local function cannot travel to a worker process
            workers = 1
    if workers <= 1 or len(jobs) < 16:
        for job in jobs:
            _filter_one(job)
        return
    pool = JobPool(_filter_one, min(workers, len(jobs)))
    for job in jobs:
        pool.submit(job, in_flight=8 * workers)
    pool.close()


WORKFLOW_TREE = ("1_WGAN/Output_Images", "1_WGAN/Models",
                 "2_CycleGAN/data/testA", "2_CycleGAN/data/testB", "2_CycleGAN/data/trainA", "2_CycleGAN/data/trainB",
                 "2_CycleGAN/generate_images/A", "2_CycleGAN/generate_images/B", "2_CycleGAN/generate_images/Synthetic_Masks_Filtered",
                 "2_CycleGAN/images", "2_CycleGAN/Models", "3_UNet/Models")


def initialize_directories(root_dir, output_dir_cyclegan, output_dir_unet):
    """The workflow's on-disk tree under ``root_dir`` plus the two result directories (HelperFunctions.py:188-238)."""
    for rel in WORKFLOW_TREE:
        os.makedirs(os.path.join(root_dir, *rel.split("/")), exist_ok=True)
    for d in (output_dir_cyclegan, output_dir_unet):
        os.makedirs(d, exist_ok=True)


class _TileSet:
    """trainA as step 0 builds it: a sink for uint8 tiles that applies the "mainly particles, not background" test
    (mean >= 1.1 x image mean for dark backgrounds, <= 0.9 x for bright ones; HelperFunctions.py:256-257,281-282)."""

    def __init__(self, directory, dark_background):
        self.directory, self.dark = directory, dark_background

    def shows_particles(self, tile, image):
        m_tile, m_img = np.mean(tile), np.mean(image)
        return m_tile >= 1.1 * m_img if self.dark else m_tile <= 0.9 * m_img

    def add(self, tile, image, source_name, tag):
        """Write ``tile`` as <source stem>-<tag><ext> if it passes the filter; returns whether it did."""
        if not self.shows_particles(tile, image):
            return False
        from PIL import Image
        ext = os.path.splitext(source_name)[-1]
        Image.fromarray(np.asarray(tile)[:, :, 0].astype('uint8')).save(os.path.join(self.directory, source_name.replace(ext, f'-{tag}{ext}')))
        return True

    def names(self):
        return get_image_file_paths_from_directory(self.directory)

    def __len__(self):
        return len(os.listdir(self.directory))


def _random_crop(image, tile_h, tile_w):
    """One augmentation sample: a random window, mirrored left-right / up-down with probability 1/2 each.  Draw order (python
    ``random``): row, column, lr coin, ud coin -- the reference's (HelperFunctions.py:272-279), so seeded runs write the same files."""
    top = random.randint(0, image.shape[0] - tile_h - 1)
    left = random.randint(0, image.shape[1] - tile_w - 1)
    crop = image[top:top + tile_h, left:left + tile_w]
    if random.random() > 0.5:
        crop = np.fliplr(crop)
    if random.random() > 0.5:
        crop = np.flipud(crop)
    return crop


def prepare_images_cycle_gan(root_dir, input_dir_images, tile_size_w=384, tile_size_h=384, num_simulated_masks=1000, dark_background=True):
    """Step 0 of the workflow (HelperFunctions.py:241-287), in three phases: (1) every SEM image is cut into a regular grid of tiles and
    the tiles that show particles go to ``2_CycleGAN/data/trainA``; (2) five random tiles are copied to ``testA``; (3) the set is
    topped up to ``num_simulated_masks`` with random, randomly mirrored crops that pass the same test.  Bit-identical files for a
    seeded ``random`` (tests/test_wgan_cpu.py)."""
    from shutil import copy
    data = os.path.join(root_dir, '2_CycleGAN', 'data')
    tiles = _TileSet(os.path.join(data, 'trainA'), dark_background)
    images = load_and_preprocess_images(input_dir_or_filelist=input_dir_images, normalization_range=None, output_channels=1)
    sources = [os.path.split(p)[-1] for p in get_image_file_paths_from_directory(input_dir_images)]

    for name, image in zip(sources, images):
        grid = np.asarray(tile_image(image, tile_size_w, tile_size_h, normalization_range=(0, 255), min_overlap=0), dtype='uint8')
        for j, tile in enumerate(grid):
            tiles.add(tile, image, name, j)

    tile_names = tiles.names()
    for path in random.sample(tile_names, 5):
        copy(path if os.path.isabs(path) else os.path.join(tiles.directory, path), os.path.join(data, 'testA'))

    wanted = num_simulated_masks - len(tiles)
    made = 0
    while made < wanted:
        # the reference names an augmented crop after tile_names[r] with r drawn over the INPUT images (HelperFunctions.py:268-271:
        # its file list had been re-bound to the tile list by then); kept, the names are part of the on-disk contract
        r = random.randint(0, images.shape[0] - 1)
        crop = _random_crop(images[r], tile_size_h, tile_size_w)
        if tiles.add(crop, images[r], os.path.split(tile_names[r])[-1], f'aug_{made}'):
            made += 1


def usable_cores():
    """CPUs this process may really use: the affinity mask AND the cgroup CPU quota (a container on a 256-thread host is typically
    given a quota -- 16 CPUs on the MI355X boxes this was measured on -- that ``os.cpu_count()`` does not show; more worker processes
    than that only get throttled)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:                                                   # cgroup v2: "<quota> <period>" or "max <period>"
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def default_workers(limit=32):
    """Worker processes for the independent host jobs of the workflow (mask placement, mask filtering, the scoring sweep): the usable
    cores less one for the dispatching process, at most ``limit``."""
    return max(1, min(limit, usable_cores() - 1))


class JobPool:
    """Worker PROCESSES (spawn context) for the independent host jobs of the workflow, with loss detection: a worker that
    dies (OOM kill under a cgroup limit) makes ``concurrent.futures`` raise BrokenProcessPool instead of blocking forever as
    ``multiprocessing.Pool`` does -- the jobs not yet finished are then run inline by the caller's process, as is everything submitted
    afterwards; so is everything when the pool cannot start (spawn re-imports ``__main__``: a caller's script without an
    ``if __name__ == "__main__":`` guard fails there).  Jobs must be idempotent (they write their own output files).  workers <= 1:
    inline from the start."""

    def __init__(self, fn, workers):
        self.fn, self.pending, self.results, self.ex = fn, [], [], None
        if workers > 1:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            try:
                self.ex = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))
            except (OSError, ValueError):
                self.ex = None

    def _inline_rest(self, why):
        import warnings
        warnings.warn(f"worker processes lost ({why!r}): the remaining jobs run in this process", stacklevel=3)
        ex, self.ex = self.ex, None
        try:
            ex.shutdown(wait=False, cancel_futures=True)
        except Exception:          # noqa: BLE001
            pass
        rest, self.pending = self.pending, []
        for _, job in rest:
            self.results.append(self.fn(job))

    def submit(self, job, in_flight=None):
        """Queue one job; in_flight: block until at most that many are unfinished (bounds what the queue holds)."""
        from concurrent.futures.process import BrokenProcessPool
        if self.ex is None:
            self.results.append(self.fn(job))
            return
        try:
            self.pending.append((self.ex.submit(self.fn, job), job))
        except (BrokenProcessPool, RuntimeError) as e:
            self.pending.append((None, job))
            self._inline_rest(e)
            return
        if in_flight is not None:
            self.drain(in_flight)

    def drain(self, keep=0):
        from concurrent.futures.process import BrokenProcessPool
        while self.ex is not None and len(self.pending) > keep:
            fut, job = self.pending[0]
            try:
                self.results.append(fut.result())
                self.pending.pop(0)
            except BrokenProcessPool as e:
                self._inline_rest(e)

    def close(self):
        self.drain(0)
        if self.ex is not None:
            self.ex.shutdown(wait=True)
            self.ex = None
        return self.results


def _jobpool_selftest_job(arg):
    """Test hook of JobPool (tests/test_host_cpu.py): "die" kills the WORKER process it runs in (what an OOM kill does); inline it survives."""
    import multiprocessing as mp
    if arg == "die":
        if mp.current_process().name != "MainProcess":
            os._exit(1)
        return "survived"
    return arg * 2


def prefetch(fetch, keys, depth=4, workers=2):
    """``fetch(k) for k in keys``, in order, with up to ``depth`` results being prepared ahead in ``workers`` threads.  The on-demand
    loaders of both trainers (``USE_DATALOADER``: PIL decode + percentile normalisation per batch, CycleGAN.py:454-479,
    UNet_Segmentation.py:104-121) are host work that the reference does between two train steps; a train step here ends with a
    device->host read of its metrics, so without this the GPU idles while a batch is decoded and the host idles while the GPU works.
    PIL, numpy and the ctypes launches release the GIL.  Order and contents are those of the plain loop (``fetch`` must not depend on
    call order: loaders that draw random numbers per batch are not prefetched by the callers).  depth <= 0: the plain loop."""
    keys = list(keys)
    if depth <= 0 or workers <= 0 or len(keys) < 2:
        for k in keys:
            yield fetch(k)
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as pool:
        pending = deque()
        it = iter(keys)
        for k in it:
            pending.append(pool.submit(fetch, k))
            if len(pending) >= depth:
                break
        try:
            while pending:
                res = pending.popleft().result()
                for k in it:
                    pending.append(pool.submit(fetch, k))
                    break
                yield res
        finally:          # the consumer stopped early (break / exception): do not prepare what nobody will read
            for f in pending:
                f.cancel()


PREFETCH_DEPTH = int(os.environ.get("SS_PREFETCH", "4"))          # batches prepared ahead by the on-demand loaders (0: off)
