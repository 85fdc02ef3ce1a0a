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
