"""Host-side reporting behind `Model.vis_batch` / `compile_batch_vis` (nlt/models/nlt.py:207-342): off the timed path,
float32 NumPy like the reference (it calls `.numpy()` on float32 tensors and hands them to xiuminglib), so the bytes that
land in the PNGs are the reference's.  What each piece stands in for:

  linear2srgb     xiuminglib/img.py:635-667   (float32 in, float32 out; np.power on the non-linear part only)
  write_arr       xiuminglib/io/img.py:36-85  (assert in [0,1]; arr * 255 TRUNCATED to uint8; PIL PNG)
  make_apng       xiuminglib/vis/video.py:15-94
  Page            xiuminglib/vis/html.py      (one table; text / image cells with captions)
  write_json      nlt/util/io.py:127-133      (indent 4, sorted keys)
  write_frames    nlt/util/io.py:90-105 -> xm.vis.video.make_video (matplotlib + ffmpeg; neither is in this image): an
                  animated PNG of the frames, and the mp4 as well wherever matplotlib's ffmpeg writer exists.
"""
import json
import os
from os.path import dirname, exists, join

import numpy as np

SRGB_LINEAR_THRES = 0.0031308
SRGB_LINEAR_COEFF = 12.92
SRGB_EXP_COEFF = 1.055
SRGB_EXPONENT = 2.4


def _mkparent(path):
    d = dirname(path)
    if d and not exists(d):
        os.makedirs(d, exist_ok=True)


def to_numpy(x):
    """torch tensor (any device) / ndarray -> ndarray, dtype kept."""
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, 'detach'):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def to_str(x):
    """A sample id as the reference decodes it (`x.numpy().decode()`): bytes, str, or a 0-d array / tensor of either."""
    if hasattr(x, 'numpy') and not isinstance(x, np.ndarray):
        x = x.numpy()
    if isinstance(x, np.ndarray):
        x = x.item()
    return x.decode() if isinstance(x, (bytes, bytearray)) else str(x)


def _check_float_0to1(arr):
    if arr.dtype.kind != 'f':
        raise TypeError("Input must be float (is %s)" % arr.dtype)
    if (arr < 0).any() or (arr > 1).any():
        raise ValueError("Input image has pixels outside [0, 1]")


def linear2srgb(im):
    if im.ndim != 3:
        raise ValueError("Input image is not even 3D (H-by-W-by-3)")
    if im.shape[2] != 3:
        raise ValueError("Input image must have 3 channels, but has %d" % im.shape[2])
    _check_float_0to1(im)
    out = im.copy()
    hi = out > SRGB_LINEAR_THRES
    lo = ~hi & (out <= SRGB_LINEAR_THRES)              # (NaN belongs to neither branch, as in the reference)
    out[lo] = out[lo] * SRGB_LINEAR_COEFF
    out[hi] = SRGB_EXP_COEFF * np.power(out[hi], 1 / SRGB_EXPONENT) - (SRGB_EXP_COEFF - 1)
    return out


def write_img(arr_uint, outpath):
    from PIL import Image
    if arr_uint.ndim == 3 and arr_uint.shape[2] == 1:
        arr_uint = np.dstack([arr_uint] * 3)
    _mkparent(outpath)
    with open(outpath, 'wb') as h:
        Image.fromarray(arr_uint).save(h, format='PNG')


def write_arr(arr_0to1, outpath, img_dtype='uint8'):
    assert arr_0to1.min() >= 0 and arr_0to1.max() <= 1, "Input should be in [0, 1], or allow it to be clipped"
    img = (arr_0to1 * np.iinfo(img_dtype).max).astype(img_dtype)
    write_img(img, outpath)
    return img


def _font(size):
    from PIL import ImageFont
    for name in ('OpenSans-Regular.ttf', 'DejaVuSans.ttf'):      # xiuminglib ships Open Sans; any TrueType PIL can find will do
        try:
            return ImageFont.truetype(name, size)
        except OSError:
            pass
    try:
        return ImageFont.load_default(size)
    except TypeError:                                            # (Pillow < 10.1: fixed-size bitmap font)
        return ImageFont.load_default()


def make_apng(imgs, labels=None, label_top_left_xy=(100, 100), font_size=100, font_color=(1, 0, 0), duration=1, outpath=None):
    """uint arrays ([H,W] / [H,W,1] / [H,W,3]) or image paths -> one animated PNG, `duration` seconds per frame."""
    from PIL import Image, ImageDraw
    if not outpath.endswith('.apng'):
        outpath += '.apng'
    _mkparent(outpath)
    font = _font(max(int(font_size), 1)) if labels is not None else None
    frames = []
    for i, img in enumerate(imgs):
        if isinstance(img, str):
            with open(img, 'rb') as h:
                pil = Image.open(h)
                pil.load()
        elif isinstance(img, np.ndarray):
            assert np.issubdtype(img.dtype, np.unsignedinteger), "If image is provided as an array, it has to be `uint`"
            if img.ndim == 2 or (img.ndim == 3 and img.shape[2] == 1):
                img = np.dstack([img.reshape(img.shape[:2])] * 3)
            pil = Image.fromarray(img)
        else:
            raise TypeError(type(img))
        if labels is not None:
            top = np.iinfo(np.array(pil).dtype).max
            ImageDraw.Draw(pil).text(tuple(label_top_left_xy), labels[i], fill=tuple(int(c * top) for c in font_color), font=font)
        frames.append(pil)
    with open(outpath, 'wb') as h:
        frames[0].save(h, format='PNG', save_all=True, append_images=frames[1:], duration=duration * 1000)
    return outpath


class Page:
    """A results page: header lines and ONE table whose rows are lists of (kind, content, caption) cells."""
    CELL = '<td align="center" valign="middle">'

    def __init__(self, title="Results", bgcolor='black', text_font='roboto', text_color='white'):
        self.title, self.bgcolor, self.text_font, self.text_color = title, bgcolor, text_font, text_color
        self.headers, self.rows = [], []

    def add_header(self, text, level=1):
        self.headers.append((level, text))

    def add_row(self, media, types, captions=None, media_width=256):
        captions = [None] * len(media) if captions is None else captions
        cells = []
        for x, kind, cap in zip(media, types, captions):
            kind = kind.lower()
            if kind == 'image':
                cell = '%s<img src="%s" alt="%s" width="%d">' % (self.CELL, x, x, media_width)
            elif kind == 'text':
                cell = '%s<p width="%d">%s</p>' % (self.CELL, media_width, x)
            else:
                raise NotImplementedError(kind)
            if cap is not None:
                cell += '\n            <br><p>%s</p>' % cap
            cells.append('            ' + cell + '</td>')
        self.rows.append('        <tr>\n' + '\n'.join(cells) + '\n        </tr>')

    def render(self, width='100%', border=6):
        head = ('<!DOCTYPE html>\n<html>\n<head>\n    <title>%s</title>\n</head>\n<body bgcolor="%s">\n<font face="%s" color="%s">\n'
                % (self.title, self.bgcolor, self.text_font, self.text_color))
        body = ''.join('    <h%d>%s</h%d>\n' % (lv, t, lv) for lv, t in self.headers)
        body += '    <table style="width:%s" border="%d">\n' % (width, border) + '\n'.join(self.rows) + '\n    </table>\n'
        return head + body + '</font>\n</body>\n</html>\n'

    def save(self, index_file):
        if not index_file.endswith('.html'):
            index_file += '.html'
        _mkparent(index_file)
        with open(index_file, 'w') as h:
            h.write(self.render())
        return index_file


def write_json(data, path):
    _mkparent(path)
    with open(path, 'w') as h:
        json.dump(data, h, indent=4, sort_keys=True)


def read_json(path):
    with open(path) as h:
        return json.load(h)


def write_frames(frames, out_mp4, fps=12):
    """The test-mode roll-up: frames (uint8 [H,W,3|4]) in the given order.  Always writes `<out>.apng` (every viewer that
    takes PNG shows its first frame) and `<out>.frames.json`; writes the .mp4 itself only where matplotlib + ffmpeg exist
    (the reference's encoder).  Returns the list of files written."""
    assert frames, "No image"
    frames = [f[:, :, :3] if f.ndim == 3 and f.shape[2] == 4 else f for f in frames]
    stem = out_mp4[:-len('.mp4')] if out_mp4.endswith('.mp4') else out_mp4
    written = [make_apng(frames, duration=1. / fps, outpath=stem + '.apng')]
    try:
        import matplotlib
        matplotlib.use('Agg')
        from matplotlib import animation, pyplot as plt
        if not animation.writers.is_available('ffmpeg'):
            raise RuntimeError('no ffmpeg')
        h, w = frames[0].shape[:2]
        fig = plt.figure(figsize=(w / 96., h / 96.), dpi=96)
        ax = fig.add_axes([0, 0, 1, 1])
        ax.axis('off')
        im = ax.imshow(frames[0], cmap='gray')
        writer = animation.writers['ffmpeg'](fps=fps)
        _mkparent(out_mp4)
        with writer.saving(fig, out_mp4, 96):
            for f in frames:
                im.set_data(f)
                writer.grab_frame()
        plt.close(fig)
        written.append(out_mp4)
    except Exception as e:                                       # noqa: BLE001 -- no encoder here: the .apng is the roll-up
        import warnings
        warnings.warn("no .mp4 written (%s: %s); the roll-up is %s" % (type(e).__name__, e, written[0]))
    return written
