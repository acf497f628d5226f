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
    """utils.py:195-207: tile [N,h,w,c] into a size[0] x size[1] grid."""
    h, w = images.shape[1], images.shape[2]
    if images.shape[3] in (3, 4):
        c = images.shape[3]
        img = np.zeros((h * size[0], w * size[1], c))
        for idx, image in enumerate(images):
            i, j = idx % size[1], idx // size[1]
            img[j * h:j * h + h, i * w:i * w + w, :] = image
        return img
    if images.shape[3] == 1:
        img = np.zeros((h * size[0], w * size[1]))
        for idx, image in enumerate(images):
            i, j = idx % size[1], idx // size[1]
            img[j * h:j * h + h, i * w:i * w + w] = image[:, :, 0]
        return img
    raise ValueError('in merge(images,size) images parameter must have dimensions: HxW or HxWx3 or HxWx4')


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
