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

__all__ = ['Cutout', 'CutoutCatalog', 'PackedImages', 'NoOverlapError', 'PartialOverlapError', 'pack_cutouts',
           'pack_cutouts_var', 'segment_bounding_boxes', 'primary_cutout_boxes']


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


def _item_tables(shapes):
    """(offsets int64 [N], total) of items packed back to back; ``shapes`` int [N, 2] = (h, w)."""
    sizes = shapes[:, 0].astype(np.int64) * shapes[:, 1].astype(np.int64)
    offs = np.zeros(len(sizes), dtype=np.int64)
    if len(sizes) > 1:
        np.cumsum(sizes[:-1], out=offs[1:])
    return offs, int(sizes.sum())


def pack_cutouts_var(frame, boxes, mask=None, fill=0.0, segmentation_image=None, ids=None, _tables=None):
    """Gather windows of DIFFERENT shapes of ``frame`` into one packed float32 device buffer
    (``spx_gather_cutouts_var_f32``): item ``k`` = ``boxes[k] = (x0, y0, width, height)`` occupies
    ``packed[offsets[k] : offsets[k] + height * width]``, row-major.  ``fill``, ``mask``,
    ``segmentation_image`` / ``ids`` as :func:`pack_cutouts`.

    Returns ``(packed, offsets, shapes)``: CUDA tensors (float32 [total], int64 [N], int32 [N, 2] = (h, w))."""
    f = device.to_device(frame, torch.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.int32)
    if f.dim() != 2 or boxes.ndim != 2 or boxes.shape[1] != 4:
        raise ValueError("frame must be 2-D and boxes [N, 4].")
    if (boxes[:, 2:] < 1).any():
        raise ValueError("Ill-formed extraction box: width and height must be positive.")
    m = None if mask is None else device.to_device(mask, torch.uint8)
    if m is not None and tuple(m.shape) != tuple(f.shape):
        raise ValueError("mask must have the shape of the frame.")
    if (segmentation_image is None) != (ids is None):
        raise ValueError("segmentation_image and ids must be given together.")
    sg = None if segmentation_image is None else device.to_device(segmentation_image, torch.int32)
    si = None if ids is None else device.to_device(ids if isinstance(ids, torch.Tensor) else np.asarray(ids, dtype=np.int32),
                                                   torch.int32)
    if _tables is None:
        shapes = np.ascontiguousarray(boxes[:, [3, 2]])
        offs, total = _item_tables(shapes)
        _tables = (torch.from_numpy(boxes).to(f.device), torch.from_numpy(offs).to(f.device),
                   torch.from_numpy(shapes).to(f.device), total)
    b_d, o_d, s_d, total = _tables
    packed = torch.empty((max(total, 1),), dtype=torch.float32, device=f.device)
    lib = _ffi.load()
    with torch.cuda.device(f.device):
        _ffi.check(lib.spx_gather_cutouts_var_f32(device.ptr(f), device.ptr(m), f.shape[0], f.shape[1],
                                                  device.ptr(b_d), boxes.shape[0], device.ptr(o_d), float(fill),
                                                  device.ptr(packed), device.ptr(sg), device.ptr(si),
                                                  device.stream_ptr()))
    return packed, o_d, s_d


class PackedImages(object):
    """Read-only sequence of 2-D float32 images stored back to back in one device buffer (what the
    catalog path of ``find_linear_fit`` returns for the interlaced cross-correlation images and the
    non-shifted blots).  ``images[k]`` is a numpy array; the buffer is copied to the host once, on first
    access -- a caller that never looks at them (the fit itself does not) pays nothing."""

    def __init__(self, buf, offsets, shapes, scale=1, stride=1, part=0):
        self._buf, self._host = buf, None
        self._off = np.asarray(offsets, dtype=np.int64) * scale
        self._shp = np.asarray(shapes, dtype=np.int64)
        self._stride, self._part = stride, part         # blots: 4 images per item, `part` selects one

    def __len__(self):
        return len(self._off)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if self._host is None:
            self._host = self._buf.cpu().numpy()
        k = int(k) + (len(self) if k < 0 else 0)
        h, w = self._shp[k]
        o = self._off[k] + self._part * h * w
        return self._host[o:o + h * w].reshape(h, w)

    def __iter__(self):
        return (self[k] for k in range(len(self)))


class CutoutCatalog(object):
    """All cutouts of ONE image for a catalog of sources: the frame stays on the device and the cutouts are
    described by an ``[N, 4]`` box table instead of N Python objects holding N array copies.

    It is a sequence of :class:`Cutout` -- ``catalog[k]`` builds the k-th one on demand with the reference's
    constructor semantics (cutout.py:689-797, ``mode='fill'``) -- so code written against lists of cutouts
    keeps working, while :func:`subpixal_amd.align.find_linear_fit` recognises a catalog and runs the whole
    loop of align.py:656-699 on the device (one gather, one blot and one cross-correlation launch per
    kernel family for all sources).

    frame : 2-D array (numpy or CUDA tensor).   boxes : int ``[N, 4]`` rows ``(x0, y0, width, height)``.
    src_pos : ``[N, 2]`` source positions in frame coordinates (default: box centres).
    src_weight : ``[N]`` or None.   src_id : ``[N]`` labels in ``segmentation_image`` (default 1..N).
    mask : bad-pixel booleans of the frame (True = bad).  segmentation_image : label image; pixels of a box
    carrying another label count as masked (cutout.py:190).  fillval : value of pixels outside the frame.
    """

    def __init__(self, frame, boxes, src_pos=None, src_weight=None, src_id=None, mask=None,
                 segmentation_image=None, wcs=None, fillval=np.nan, exptime=1, data_units='rate'):
        self.boxes = np.ascontiguousarray(boxes, dtype=np.int32)
        if self.boxes.ndim != 2 or self.boxes.shape[1] != 4:
            raise ValueError("boxes must have shape [N, 4].")
        if (self.boxes[:, 2:] < 1).any():
            raise ValueError("Ill-formed extraction box: width and height must be positive.")
        n = len(self.boxes)
        self.frame = device.to_device(frame, torch.float32)
        if self.frame.dim() != 2:
            raise ValueError("frame must be 2-D.")
        self._frame_host = frame if isinstance(frame, np.ndarray) else None
        self._tables = None
        self.mask = None if mask is None else device.to_device(mask, torch.uint8)
        self.segmentation_image = None if segmentation_image is None else \
            device.to_device(segmentation_image, torch.int32)
        self.wcs = wcs
        self.fillval = fillval
        self.exptime = exptime
        self.data_units = data_units
        b = self.boxes.astype(np.float64)
        self.src_pos = (np.stack([b[:, 0] + 0.5 * (b[:, 2] - 1), b[:, 1] + 0.5 * (b[:, 3] - 1)], axis=1)
                        if src_pos is None else np.asarray(src_pos, dtype=np.float64).reshape(n, 2))
        self.src_weight = None if src_weight is None else np.asarray(src_weight, dtype=np.float64).reshape(n)
        if self.src_weight is not None and (self.src_weight < 0).any():
            raise ValueError("Source weight must be a non-negative number or None.")
        self.src_id = np.arange(1, n + 1, dtype=np.int32) if src_id is None else \
            np.asarray(src_id, dtype=np.int32).reshape(n)

    def __len__(self):
        return len(self.boxes)

    @property
    def shapes(self):
        """``[N, 2]`` int32 (height, width)."""
        return np.ascontiguousarray(self.boxes[:, [3, 2]])

    def _host_frame(self):
        if self._frame_host is None:
            self._frame_host = self.frame.cpu().numpy()
        return self._frame_host

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        k = int(k) + (len(self) if k < 0 else 0)
        x0, y0, w, h = (int(v) for v in self.boxes[k])
        ct = Cutout(self._host_frame(), self.wcs, blc=(x0, y0), trc=(x0 + w - 1, y0 + h - 1),
                    src_pos=tuple(self.src_pos[k]),
                    src_weight=None if self.src_weight is None else float(self.src_weight[k]),
                    src_id=int(self.src_id[k]), data_units=self.data_units, exptime=self.exptime,
                    mode='fill', fillval=self.fillval)
        for extra, test in ((self.mask, lambda a: np.asarray(a, bool)),
                            (self.segmentation_image, lambda a: np.asarray(a) != int(self.src_id[k]))):
            if extra is not None:
                ex = extra.cpu().numpy() if isinstance(extra, torch.Tensor) else np.asarray(extra)
                ct.mask[ct.insertion_slice] |= test(ex[ct.extraction_slice])
        return ct

    def __iter__(self):
        return (self[k] for k in range(len(self)))

    def packed(self, zero_masked=False):
        """``(packed, offsets, shapes)`` device tensors of all cutouts (:func:`pack_cutouts_var`).
        zero_masked False: the cutouts' ``data`` -- frame pixels, ``fillval`` outside the frame;
        True: masked pixels (mask, other segments, non-finite, outside) zeroed, i.e. after align.py:661."""
        if self._tables is None:           # the box / offset / shape tables go to the device once
            shapes = self.shapes
            offs, total = _item_tables(shapes)
            dev = self.frame.device
            self._tables = (torch.from_numpy(self.boxes).to(dev), torch.from_numpy(offs).to(dev),
                            torch.from_numpy(shapes).to(dev), total)
            self._ids_dev = torch.from_numpy(self.src_id).to(dev)
        if zero_masked:
            return pack_cutouts_var(self.frame, self.boxes, self.mask, 0.0, self.segmentation_image,
                                    None if self.segmentation_image is None else self._ids_dev, self._tables)
        return pack_cutouts_var(self.frame, self.boxes, None, self.fillval, _tables=self._tables)
