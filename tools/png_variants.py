#!/usr/bin/env python3
"""PNG files of every colour type and bit depth, plain and Adam7-interlaced, with all five row filters, palette and
tRNS chunks: inputs for checking a decoder. Written with zlib only.

    python tools/png_variants.py <outdir>      -> writes the files, prints their names"""
import os
import struct
import sys
import zlib

import numpy as np

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def pack_row(samples, depth):
    """samples: 1-D int array of one row (all channels interleaved) -> bytes at `depth` bits per sample."""
    if depth == 8:
        return samples.astype(np.uint8).tobytes()
    if depth == 16:
        return samples.astype(">u2").tobytes()
    per = 8 // depth
    pad = (-len(samples)) % per
    s = np.concatenate([samples, np.zeros(pad, dtype=samples.dtype)]).reshape(-1, per)
    out = np.zeros(len(s), dtype=np.uint8)
    for k in range(per):
        out |= (s[:, k].astype(np.uint8) << (8 - depth * (k + 1))).astype(np.uint8)
    return out.tobytes()


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp):
    """Applies filter type (row index mod 5) to each packed row; returns the filtered stream."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = y % 5
        cur = bytearray(len(row))
        for i, x in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
            cur[i] = (x - pred) & 0xFF
        out.append(ft)
        out += cur
        prev = row
    return bytes(out)


def write_png(path, img, color_type, depth, interlace=False, palette=None, trns=None):
    """img: [h, w, channels] int array of samples in 0 .. 2**depth - 1 (palette indices for colour type 3)."""
    h, w, ch = img.shape
    assert ch == CHANNELS[color_type]
    bpp = max(1, ch * depth // 8)
    stream = b""
    passes = ADAM7 if interlace else [(0, 0, 1, 1)]
    for x0, y0, dx, dy in passes:
        sub = img[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = [pack_row(sub[y].reshape(-1), depth) for y in range(sub.shape[0])]
        stream += filter_rows(rows, bpp)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    z = zlib.compress(stream, 6)
    half = len(z) // 2
    data += chunk(b"IDAT", z[:half]) + chunk(b"tEXt", b"Comment\x00two IDAT chunks") + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")
    open(path, "wb").write(data)


def write_all(out):
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(42)
    names = []
    for color_type, depths in ((0, (1, 2, 4, 8, 16)), (2, (8, 16)), (3, (1, 2, 4, 8)), (4, (8, 16)), (6, (8, 16))):
        for depth in depths:
            for interlace in (False, True):
                for w, h in ((13, 9), (1, 1), (5, 17)):
                    ch = CHANNELS[color_type]
                    img = rng.integers(0, 2 ** depth, size=(h, w, ch))
                    palette = trns = None
                    if color_type == 3:
                        n = 2 ** depth
                        palette = rng.integers(0, 256, size=n * 3).tolist()
                        trns = rng.integers(0, 256, size=max(1, n // 2)).tolist()
                    elif color_type == 0 and depth in (2, 16):
                        key = int(img[0, 0, 0])
                        trns = [key >> 8, key & 255]
                    elif color_type == 2 and depth == 8:
                        k = img[0, 0]
                        trns = [0, int(k[0]), 0, int(k[1]), 0, int(k[2])]
                    name = f"ct{color_type}_d{depth}_{'adam7' if interlace else 'plain'}_{w}x{h}.png"
                    write_png(os.path.join(out, name), img, color_type, depth, interlace, palette, trns)
                    names.append(name)
    return names


if __name__ == "__main__":
    for n in write_all(sys.argv[1]):
        print(n)
