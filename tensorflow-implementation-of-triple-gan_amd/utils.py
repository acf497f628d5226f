"""Sample-grid utilities — the pure-NumPy part of the reference's utils.py that the hot path's artefacts need:
save_images / merge / image_manifold_size / inverse_transform (utils.py:192-231).  scipy.misc.imsave
(utils.py:209) no longer exists; PNGs are written with zlib directly."""
import struct
import zlib

import numpy as np


def image_manifold_size(num_images):
    """utils.py:224-231."""
    manifold_h = int(np.floor(np.sqrt(num_images)))
    manifold_w = int(np.ceil(np.sqrt(num_images)))
    assert manifold_h * manifold_w == num_images
    return manifold_h, manifold_w


def inverse_transform(images):
    """utils.py:221-222: [-1,1] -> [0,1]."""
    return (images + 1.) / 2.


def merge(images, size):
    """Tile [N,h,w,c] (c in 1, 3, 4) row-major into a size[0] x size[1] grid — the result of utils.py:195-207, as one reshape /
    transpose; grid cells beyond N stay zero; single-channel grids come back 2-D."""
    images = np.asarray(images)
    n, h, w, c = images.shape
    if c not in (1, 3, 4):
        raise ValueError('merge: images must be [N,H,W,1], [N,H,W,3] or [N,H,W,4], got %r' % (images.shape,))
    rows, cols = int(size[0]), int(size[1])
    cells = np.zeros((rows * cols, h, w, c), np.float64)
    cells[:min(n, rows * cols)] = images[:rows * cols]
    grid = cells.reshape(rows, cols, h, w, c).transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, c)
    return grid[:, :, 0] if c == 1 else grid


def write_png(path, img):
    """img: float [H,W] or [H,W,3] in [0,1] -> 8-bit PNG."""
    a = np.clip(np.asarray(img) * 255.0 + 0.5, 0, 255).astype(np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    color = {1: 0, 3: 2, 4: 6}[c]
    raw = b''.join(b'\x00' + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, color, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def save_images(images, size, image_path):
    """utils.py:192-193,209-210."""
    return write_png(image_path, merge(inverse_transform(images), size))
