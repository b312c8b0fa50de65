"""Cutout carrier and device-side cutout packing.

``Cutout`` keeps the attribute names the reference's hot path reads
(/root/reference/subpixal/cutout.py:590-1157: ``data mask dq weight blc trc dx dy
width height naxis src_pos cutout_src_pos src_weight src_id wcs``) so objects of
either package can be handed to :func:`subpixal_amd.align.find_linear_fit`.  It is a
plain host-side container; WCS handling is delegated to whatever ``wcs`` object the
caller supplies (astropy is not a dependency here).

``pack_cutouts`` is the batched device counterpart of ``Cutout.__init__``'s
slicing/fill (cutout.py:737-755) plus ``align.py:661``'s masked-pixel zeroing: one
gather kernel instead of thousands of tiny host slices and copies.
"""
import numpy as np
import torch

from . import _ffi, device

__all__ = ['Cutout', 'NoOverlapError', 'PartialOverlapError', 'pack_cutouts',
           'segment_bounding_boxes', 'primary_cutout_boxes']


class NoOverlapError(ValueError):
    """The extraction box lies completely outside the image."""


class PartialOverlapError(ValueError):
    """The extraction box overhangs the image and ``mode`` is 'strict'."""


class Cutout(object):
    """A rectangular cutout ``[blc, trc]`` (both corners inclusive, ``(x, y)``) of an
    image, with a bad-pixel mask (True = bad: outside the image or non-finite).

    Parameters follow the reference class (cutout.py:689-691): ``data, wcs, blc, trc,
    src_pos, src_weight, dq, weight, src_id, data_units, exptime, mode, fillval``.
    """

    def __init__(self, data, wcs=None, blc=(0, 0), trc=None, src_pos=None,
                 src_weight=None, dq=None, weight=None, src_id=0, data_units='rate',
                 exptime=1, mode='strict', fillval=np.nan):
        if data is None and trc is None:
            raise ValueError("'trc' cannot be None when 'data' is None.")
        if mode not in ('strict', 'fill'):
            raise ValueError("Argument 'mode' must be either 'strict' or 'fill'.")
        if data is None:
            nx, ny, dtype = trc[0] + 1, trc[1] + 1, np.float32
        else:
            data = np.asarray(data)
            (ny, nx), dtype = data.shape, data.dtype
        if trc is None:
            trc = (nx - 1, ny - 1)
        if blc[0] >= nx or blc[1] >= ny or trc[0] < 0 or trc[1] < 0:
            raise NoOverlapError("Cutout's extraction box does not overlap image data array.")
        if trc[0] < blc[0] or trc[1] < blc[1]:
            raise ValueError("Ill-formed extraction box: coordinates of the top-right corner "
                             "cannot be smaller than the coordinates of the bottom-left corner.")
        if mode == 'strict' and (blc[0] < 0 or blc[1] < 0 or trc[0] >= nx or trc[1] >= ny):
            raise PartialOverlapError("Cutout's extraction box only partially overlaps image "
                                      "data array.")
        self._blc = (int(blc[0]), int(blc[1]))
        self._trc = (int(trc[0]), int(trc[1]))
        self.src_pos = src_pos
        self.src_weight = src_weight
        self.src_id = src_id
        self.dx = 0
        self.dy = 0
        self.wcs = wcs
        self.exptime = exptime
        self.data_units = data_units

        # overlap of the box with the image, in image and in cutout coordinates
        x1, y1 = max(0, blc[0]), max(0, blc[1])
        x2, y2 = min(nx - 1, trc[0]) + 1, min(ny - 1, trc[1]) + 1
        self.extraction_slice = np.s_[y1:y2, x1:x2]
        self.insertion_slice = np.s_[y1 - blc[1]:y2 - blc[1], x1 - blc[0]:x2 - blc[0]]

        shape = (self.height, self.width)
        self._data = np.full(shape, fillval, dtype=dtype)
        self._mask = np.ones(shape, dtype=bool)
        if data is not None:
            self._data[self.insertion_slice] = data[self.extraction_slice]
        self._mask[self.insertion_slice] = False
        self._mask |= ~np.isfinite(self._data)
        self._dq = self._crop(dq, (ny, nx), 'DQ')
        self._weight = self._crop(weight, (ny, nx), 'weight')

    def _crop(self, arr, shape, what):
        if arr is None:
            return None
        arr = np.asarray(arr)
        if arr.shape != shape:
            raise ValueError("Image's %s array shape must match the shape of image 'data'." % what)
        out = np.zeros((self.height, self.width), dtype=arr.dtype)
        out[self.insertion_slice] = arr[self.extraction_slice]
        return out

    # geometry ------------------------------------------------------------
    @property
    def blc(self):
        return self._blc

    @blc.setter
    def blc(self, v):
        self._blc = (v[0], v[1])

    @property
    def trc(self):
        return self._trc

    @trc.setter
    def trc(self, v):
        self._trc = (v[0], v[1])

    @property
    def width(self):
        return self._trc[0] - self._blc[0] + 1

    @property
    def height(self):
        return self._trc[1] - self._blc[1] + 1

    @property
    def naxis(self):
        return [self.width, self.height]

    @property
    def naxis1(self):
        return self.width

    @property
    def naxis2(self):
        return self.height

    # source --------------------------------------------------------------
    @property
    def src_pos(self):
        """Source position in the coordinates of the image the cutout came from."""
        return self._src_pos

    @src_pos.setter
    def src_pos(self, pos):
        if pos is None:
            self._src_pos = (0.5 * (self._blc[0] + self._trc[0]), 0.5 * (self._blc[1] + self._trc[1]))
        else:
            self._src_pos = tuple(pos)[:2]

    @property
    def cutout_src_pos(self):
        """Source position in the cutout's own pixel coordinates."""
        return (self._src_pos[0] - self._blc[0], self._src_pos[1] - self._blc[1])

    @cutout_src_pos.setter
    def cutout_src_pos(self, pos):
        self.src_pos = None if pos is None else (pos[0] + self._blc[0], pos[1] + self._blc[1])

    @property
    def src_weight(self):
        return self._src_weight

    @src_weight.setter
    def src_weight(self, w):
        if w is not None and np.any(np.asarray(w) < 0.0):
            raise ValueError("Source weight must be a non-negative number or None.")
        self._src_weight = w

    @property
    def exptime(self):
        return self._exptime

    @exptime.setter
    def exptime(self, t):
        if t <= 0:
            raise ValueError("'exptime' must be positive.")
        self._exptime = t

    @property
    def data_units(self):
        return self._data_units

    @data_units.setter
    def data_units(self, units):
        units = units.lower()
        if units not in ('rate', 'counts'):
            raise ValueError("Allowed image data units are: 'rate' or 'counts'.")
        self._data_units = units

    # arrays --------------------------------------------------------------
    def _same_shape(self, arr, name):
        arr = np.asarray(arr)
        if arr.shape != self._data.shape:
            raise ValueError("could not broadcast input array from shape (%s) into shape (%s)"
                             % (','.join(map(str, arr.shape)), ','.join(map(str, self._data.shape))))
        return arr

    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, d):
        if d is None:
            raise ValueError("'data' cannot be None.")
        if d is not self._data:
            self._data = self._same_shape(d, 'data')

    @property
    def mask(self):
        return self._mask

    @mask.setter
    def mask(self, m):
        if m is None:
            raise ValueError("'mask' cannot be None.")
        if m is not self._mask:
            self._mask = self._same_shape(np.asarray(m, dtype=bool), 'mask')

    @property
    def dq(self):
        return self._dq

    @dq.setter
    def dq(self, d):
        self._dq = None if d is None else self._same_shape(d, 'dq')

    @property
    def weight(self):
        return self._weight

    @weight.setter
    def weight(self, w):
        self._weight = None if w is None else self._same_shape(w, 'weight')

    # coordinates (need a wcs object with all_pix2world / all_world2pix) ----
    def pix2world(self, x, y, origin=0):
        if self.wcs is None:
            raise ValueError("WCS was not set.")
        x = np.asarray(x, dtype=np.float64) + (self._blc[0] - self.dx)
        y = np.asarray(y, dtype=np.float64) + (self._blc[1] - self.dy)
        return list(self.wcs.all_pix2world(x, y, origin))

    def world2pix(self, ra, dec, origin=0):
        if self.wcs is None:
            raise ValueError("WCS was not set.")
        x, y = self.wcs.all_world2pix(np.asarray(ra, np.float64), np.asarray(dec, np.float64), origin)
        return [x - (self._blc[0] - self.dx), y - (self._blc[1] - self.dy)]


def segment_bounding_boxes(segmentation_image, max_label=None):
    """Bounding box and pixel count of every label of a segmentation image in one GPU pass
    (the reference scans the whole frame once per source, cutout.py:151-160).

    Returns ``boxes [max_label + 1, 4]`` int32 = (xmin, ymin, xmax, ymax) inclusive and
    ``counts [max_label + 1]`` as CUDA tensors; row 0 (background) and absent labels hold
    (INT32_MAX, INT32_MAX, -1, -1) / 0."""
    seg = device.to_device(segmentation_image, torch.int32)
    if seg.dim() != 2:
        raise ValueError("segmentation image must be 2-D.")
    if max_label is None:
        max_label = int(seg.max().item()) if seg.numel() else 0
    boxes = torch.empty((max_label + 1, 4), dtype=torch.int32, device=seg.device)
    counts = torch.empty((max_label + 1,), dtype=torch.int32, device=seg.device)
    lib = _ffi.load()
    with torch.cuda.device(seg.device):
        _ffi.check(lib.spx_label_bboxes_i32(device.ptr(seg), seg.shape[0], seg.shape[1], int(max_label),
                                            device.ptr(boxes), device.ptr(counts), device.stream_ptr()))
    return boxes, counts


def primary_cutout_boxes(segmentation_image, ids=None, pad=1):
    """Extraction boxes of the primary cutouts, with the reference's rules
    (``create_primary_cutouts``, cutout.py:138-175): only labels present in the image (and in
    ``ids`` when given), sources whose segment touches the image border are skipped
    (cutout.py:162-167), the box is the segment's bounding rectangle grown by ``pad``
    (rounded away from zero, cutout.py:139).

    Returns ``(kept_ids, boxes)``: numpy int32 arrays, boxes rows ``(x0, y0, width, height)``
    as :func:`pack_cutouts` takes them."""
    ny, nx = segmentation_image.shape
    pad = int(np.ceil(pad)) if pad >= 0 else int(np.floor(pad))
    bb, cnt = segment_bounding_boxes(segmentation_image)
    bb = bb.cpu().numpy()
    cnt = cnt.cpu().numpy()
    present = np.nonzero(cnt[1:] > 0)[0] + 1
    if ids is not None:
        present = np.intersect1d(np.asarray(ids), present)
    b = bb[present]
    inside = (b[:, 0] > 0) & (b[:, 1] > 0) & (b[:, 2] < nx - 1) & (b[:, 3] < ny - 1)
    present, b = present[inside], b[inside]
    boxes = np.stack([b[:, 0] - pad, b[:, 1] - pad, b[:, 2] - b[:, 0] + 1 + 2 * pad,
                      b[:, 3] - b[:, 1] + 1 + 2 * pad], axis=1).astype(np.int32)
    return present.astype(np.int32), boxes


def pack_cutouts(frame, boxes, tile, mask=None, fill=0.0, segmentation_image=None, ids=None):
    """Gather ``len(boxes)`` windows of ``frame [fny, fnx]`` into ``tiles [N, tny, tnx]``
    float32 on the device.

    boxes : int ``[N, 4]`` rows ``(x0, y0, width, height)``; windows may overhang the
        frame.  mask : bad-pixel booleans ``[fny, fnx]`` (True = bad) or None.
    segmentation_image, ids : optional label image ``[fny, fnx]`` and the label of each box's
        source ``[N]``; pixels carrying another label are filled as well (cutout.py:190).
    Inside its window a tile holds the frame pixel, or ``fill`` where the window leaves
    the frame, the pixel is masked or it is not finite; the padding outside the window
    is 0 (which leaves the linear cross-correlation unchanged).
    """
    f = device.to_device(frame, torch.float32)
    b = device.to_device(np.asarray(boxes, dtype=np.int32) if not isinstance(boxes, torch.Tensor)
                         else boxes, torch.int32)
    if f.dim() != 2 or b.dim() != 2 or b.shape[1] != 4:
        raise ValueError("frame must be 2-D and boxes [N, 4].")
    m = None if mask is None else device.to_device(mask, torch.uint8)
    if m is not None and tuple(m.shape) != tuple(f.shape):
        raise ValueError("mask must have the shape of the frame.")
    tny, tnx = int(tile[0]), int(tile[1])
    if int((b[:, 2] > tnx).any()) or int((b[:, 3] > tny).any()):
        raise ValueError("a box is larger than the tile.")
    if (segmentation_image is None) != (ids is None):
        raise ValueError("segmentation_image and ids must be given together.")
    sg = None if segmentation_image is None else device.to_device(segmentation_image, torch.int32)
    si = None if ids is None else device.to_device(np.asarray(ids, dtype=np.int32)
                                                   if not isinstance(ids, torch.Tensor) else ids, torch.int32)
    if sg is not None and (tuple(sg.shape) != tuple(f.shape) or si.shape[0] != b.shape[0]):
        raise ValueError("segmentation image must match the frame and ids the boxes.")
    tiles = torch.empty((b.shape[0], tny, tnx), dtype=torch.float32, device=f.device)
    lib = _ffi.load()
    with torch.cuda.device(f.device):
        _ffi.check(lib.spx_gather_cutouts_f32(device.ptr(f), device.ptr(m), f.shape[0], f.shape[1],
                                              device.ptr(b), b.shape[0], tny, tnx, float(fill),
                                              device.ptr(tiles), device.ptr(sg), device.ptr(si),
                                              device.stream_ptr()))
    return tiles
