"""Host-side image helpers with the reference's names and semantics (Releases/Version 1.2.0/HelperFunctions.py):
``get_image_file_paths_from_directory`` :290-291, ``load_and_preprocess_images`` :294-329,
``tile_image`` :17-62, ``stitch_image`` :65-141.  Pure numpy/PIL; feeds and drains the GPU path."""
import math
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


def eight_to_four_connected(img):
    """HelperFunctions.py:144-152: break diagonal-only (8-connected) contacts, scanning in the reference's order."""
    if np.count_nonzero(img) > 2 or np.count_nonzero(img) < img.size - 2:
        for x in range(0, img.shape[0] - 1):
            for y in range(0, img.shape[1] - 1):
                if img[x, y] == 0 and img[x + 1, y + 1] == 0 and img[x + 1, y] != 0 and img[x, y + 1] != 0:
                    img[x + 1, y] = 0
                elif img[x + 1, y] == 0 and img[x, y + 1] == 0 and img[x, y] != 0 and img[x + 1, y + 1] != 0:
                    img[x, y] = 0
    return img


def segment(image, threshold, watershed_lines, min_distance=9, use_four_connectivity=True):
    """HelperFunctions.py:155-160 / Measurements.py:263-305 with darkBackground=True: Otsu (threshold < 0) or fixed
    threshold, optional 8->4 connectivity.  The watershed split (skimage peak_local_max + watershed, not installed here)
    is NOT implemented: ``watershed_lines=True`` raises, callers that want the reference default must opt out explicitly."""
    img = np.asarray(image).copy()
    if threshold < 0:
        threshold = threshold_otsu(img)
    mask = img > threshold
    if watershed_lines and np.min(mask) != np.max(mask):
        raise NotImplementedError("watershed post-processing (Measurements.py:286-305) is a 'next' row (SURVEY 8f #2); "
                                  "call with watershed_lines=False for threshold-only label maps")
    labels = np.asarray(mask * 255, dtype='uint8')
    if use_four_connectivity:
        labels = eight_to_four_connected(labels)
    return labels
